"""ALADModel-compatible loss orchestration (reference alad/alad_model.py:250-454) over the HIP
criteria of aladin_amd.loss.

Scope: `forward`, `forward_loss` and the attribute protocol
(`logger`, `Eiters`, `losses_types`, `losses_weights`, `*_criterion`, `distillation_loss`) are
re-stated; the encoder (`JointTextImageTransformerEncoder`, alad_model.py:29-247) is host PyTorch
code: aladin_amd/encoder.py provides it -- matching head `final_projection_net` included -- around
an INJECTED VinVL/Oscar `backbone` (the BERT itself is out of scope), or any `encoder` module returning
the reference's 7-tuple (img_glob (B,D), cap_glob (B,D), img_set (R,B,D), cap_seq (T,B,D), img_len,
cap_len, reg_loss) can be passed.  (Random-init stand-ins for smoke tests live in tests/standins.py.)
"""
import os

import torch
from torch import nn

from . import ops
from .loss import AlignmentContrastiveLoss, ContrastiveLoss, DistillationLoss, dot_sim


class ALADModel(nn.Module):
    def __init__(self, config, oscar_checkpoint=None, encoder=None, backbone=None, shard_group=False, backbone_autocast=None,
                 graphed=None):
        """graphed: True = the loss heads of a TRAINING step (forward_loss + the weighted sum + their backward down to the encoder's
        four outputs) are captured once per batch shape into a HIP graph and replayed (aladin_amd.graphs.GraphedLossStep) -- the
        import swap of INTEGRATION.md section 2 then costs the replay (~0.07 - 0.1 ms at the shipped bs 32) instead of ~0.2 ms of
        eager launches, with NO change to the train loop: same `(loss, loss_dict)`, `loss.backward()` continues into the encoder,
        and `model.logger` is current whenever it is READ (the logged terms leave the device by one asynchronous copy per step;
        reading `model.logger` -- what alad/train.py:436,447 do -- waits for the outstanding ones first).  None (default): the
        environment variable ALADIN_GRAPH_HEADS (unset / "0" = off).  Evaluation (no_grad), sharded steps and configurations outside
        the shipped ones (`_fused_heads_ok`) run eagerly as before.  The values of the returned loss_dict live in the graph's static
        buffers: they are the current step's until the next step of the same batch shape runs (the reference's loop never keeps them).
        shard_group: False = single device (the reference, alad/train.py:251-255); None or a torch.distributed group =
        one process per GPU, the loss heads run on the GLOBAL batch (all ranks' samples) with the score matrices sharded
        by caption block (aladin_amd.distributed.sharded_loss_heads; BASELINE configs[3] for the shipped YAMLs).
        encoder: any module with the 7-tuple contract of JointTextImageTransformerEncoder; or backbone: the
        VinVL/Oscar model (anything with the `.bert(...)` call of alad_model.py:129,139), around which
        aladin_amd.encoder.JointTextImageTransformerEncoder -- matching head included -- is built here as
        the reference does at :259."""
        super().__init__()
        if encoder is None and (backbone is not None or oscar_checkpoint is not None):
            from .encoder import JointTextImageTransformerEncoder
            encoder = JointTextImageTransformerEncoder(config, backbone, oscar_checkpoint, backbone_autocast)     # :259
        self.img_txt_enc = encoder                                   # alad_model.py:259 (injected here)
        training = config['training']
        self.losses_types = training['loss-type'].split('-')          # :265
        self.losses_weights = training['loss-weights']                # :266
        if isinstance(self.losses_weights, list):                     # :267-270
            assert len(self.losses_types) == len(self.losses_weights)
            self.losses_weights = {k: v for k, v in zip(self.losses_types, self.losses_weights)}
            self.auto_weight = False
        else:                                                         # :271-273 ('auto': unregistered parameters)
            dev = 'cuda' if torch.cuda.is_available() else 'cpu'
            self.losses_weights = {k: nn.Parameter(-2.3 * torch.ones(1)).to(dev) for k in self.losses_types}
            self.auto_weight = True
        if 'distillation' in self.losses_types:                       # :275-276
            self.distillation_loss = DistillationLoss(mode=training['distillation-mode'])
        if 'attdistillation' in self.losses_types:
            raise NotImplementedError("aladin_amd: 'attdistillation' is constructed but never used by the "
                                      "reference's forward_loss (alad_model.py:278-279); not provided")
        if 'alignment' in self.losses_types or 'distillation' in self.losses_types:    # :285-288
            self.alignment_criterion = AlignmentContrastiveLoss(
                margin=training['margin'], measure=training['measure'],
                max_violation=training['max-violation'], aggregation=training['alignment-mode'])
        self.matching_criterion = ContrastiveLoss(                    # :289-292 (the `if` there is always true)
            margin=training['margin'], measure=training['measure'], max_violation=training['max-violation'])
        self.Eiters = 0
        self.config = config
        self._logger = None
        self.pending_log = None
        self.shard_group = shard_group
        self.graphed = bool(os.environ.get('ALADIN_GRAPH_HEADS', '') not in ('', '0')) if graphed is None else bool(graphed)
        self._graph_step = None
        self._graph_bypass = False                                    # set by GraphedLossStep while IT runs forward_loss_total

    @property
    def logger(self):
        """The callers' LogCollector (alad/train.py:413, alad/evaluation.py:109 swap it in).  With graphed heads the logged terms
        of the last steps may still be on their way to the host: reading the logger delivers them first."""
        if self._graph_step is not None:
            self._graph_step.flush()
        return self._logger

    @logger.setter
    def logger(self, value):
        # (alad/train.py:413 re-assigns the SAME collector every iteration: nothing to deliver then, and no host wait)
        if self._graph_step is not None and value is not self._logger:
            self._graph_step.flush()                                  # outstanding values belong to the logger that was current
        self._logger = value

    def forward_emb(self, example_imgs, example_txts):
        """reference alad_model.py:325-348 (host->device copies + encoder call)."""
        if self.img_txt_enc is None:
            raise RuntimeError('aladin_amd.ALADModel: no encoder injected (the VinVL backbone is outside the '
                               'accelerated path; pass encoder=...)')
        if torch.cuda.is_available():
            example_imgs = [c.cuda() if isinstance(c, torch.Tensor) else c for c in example_imgs]
            example_txts = [c.cuda() if isinstance(c, torch.Tensor) else c for c in example_txts]
        return self.img_txt_enc(example_imgs, example_txts)

    def forward_loss(self, img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, reg_loss, log=True):
        """reference alad_model.py:371-428.  Same terms, same insertion order, same logger keys; the
        per-term `.item()` host syncs of the reference are batched into one device->host copy.
        log=False (used under HIP-graph capture, where a host sync is illegal: aladin_amd.graphs) skips the
        logger update and leaves the (key, tensor, n) triples in `self.pending_log` for `flush_log()`.
        The length arguments may be Python lists (as the reference passes them) or int32 device tensors."""
        losses = {}
        logged = []                                                   # (key, tensor, n)
        img_emb_set = img_emb_set.permute(1, 0, 2)                    # :377-378  (S,B,D) -> (B,S,D) views
        cap_emb_seq = cap_emb_seq.permute(1, 0, 2)

        wants_matching = 'matching' in self.config['training']['loss-type']      # :381 (substring test on the string)
        wants_align = 'alignment' in self.losses_types or 'distillation' in self.losses_types
        wants_distill = 'distillation' in self.losses_types
        mc = self.matching_criterion
        sim = getattr(mc, 'sim', None)

        # The alignment head first: its score matrix is the distillation teacher.  (The reference computes the
        # matching term first, :380; the terms are independent, and `losses` / the logger are filled in the
        # reference's order below.)
        if wants_align:                                               # :385-390
            alignment_loss, teacher_scores = self.alignment_criterion(
                img_emb_set, cap_emb_seq, img_lengths, cap_lengths, return_similarity_mat=True)

        # Small batches (every shipped YAML: bs 32) with the shipped 'dot' measure and listnet distillation: matching
        # scores, their hinge and the distillation loss in ONE forward / ONE backward launch (ops.small_batch_match_distill)
        small = (img_emb.shape[0] <= ops.SMALL_BATCH_MAX and sim is dot_sim and 'selfaggregation' not in self.losses_types
                 and (not wants_distill or self.distillation_loss.mode == 'listnet') and (wants_matching or wants_distill))
        if small:
            matching_loss, distillation_loss, matching_mat = ops.small_batch_match_distill(
                img_emb, cap_emb, teacher_scores if wants_distill else None, mc.margin, mc.max_violation, want_hinge=wants_matching)
        elif wants_matching or sim is None:
            matching_loss, matching_mat = mc(img_emb, cap_emb, return_similarity_mat=True)   # :380
        elif wants_distill or 'selfaggregation' in self.losses_types:
            # the reference computes the matching hinge here and drops it (:380-381); only the score matrix
            # is used further down (distillation student, :405), so the two hinge launches are skipped
            matching_mat = sim(img_emb, cap_emb)
        if wants_matching:
            losses['matching'] = matching_loss
            logged.append(('matching_loss', matching_loss, img_emb.size(0)))

        if 'alignment' in self.losses_types:
            losses['alignment'] = alignment_loss
            logged.append(('alignment_loss', alignment_loss, img_emb_set.size(0)))

        if 'selfaggregation' in self.losses_types:                    # :397-402
            matching_loss, matching_mat = mc(img_emb, cap_emb, return_similarity_mat=True)
            losses['selfaggregation'] = matching_loss
            logged.append(('self_attention_loss', matching_loss, img_emb.size(0)))

        if wants_distill:                                             # :404-408
            if not small:
                distillation_loss = self.distillation_loss(teacher_scores, matching_mat)
            losses['distillation'] = distillation_loss
            logged.append(('distillation_loss', distillation_loss, img_emb.size(0)))

        if 'entropy' in self.losses_types:
            raise NotImplementedError("aladin_amd: the 'entropy' term (alad_model.py:410-421) is not provided")

        if 'regularizehidden' in self.losses_types:                   # :423-425
            losses['regularizehidden'] = reg_loss
            logged.append(('regularize_hidden_loss', reg_loss, img_emb.size(0)))

        self.pending_log = logged
        if log:
            self.flush_log()
        return losses

    def flush_log(self):
        """Push the loss terms of the last forward_loss(log=False) call to `self.logger` (one device->host copy)."""
        logged, self.pending_log = getattr(self, 'pending_log', None), None
        if self._logger is not None and logged:
            vals = torch.stack([torch.as_tensor(t).detach().reshape(()).to(torch.float32) for _, t, _ in logged]).tolist()
            for (key, _, n), v in zip(logged, vals):
                self._logger.update(key, v, n)

    def weighted_total(self, loss_dict, epoch=0, distill_epoch=2):
        """reference alad_model.py:442-453: drop the distillation term before `distill_epoch` (when another term
        exists), then the fixed-weight sum, or 0.5 * sum(L e^-w + w) for 'auto' weights.  Mutates loss_dict like
        the reference."""
        if epoch < distill_epoch and len(loss_dict) > 1:              # :442-444
            loss_dict.pop('distillation', None)
        loss = 0
        if self.auto_weight:                                          # :445-449
            for k in loss_dict:
                loss += loss_dict[k] * torch.exp(-self.losses_weights[k]) + self.losses_weights[k]
            loss *= 0.5
        else:                                                         # :450-453
            for k in loss_dict:
                loss += loss_dict[k] * self.losses_weights[k]
        return loss

    def _fused_heads_ok(self, img_emb):
        """The single-node step (ops.loss_heads) covers what every shipped YAML trains with: measure 'dot', 'MrSw'
        alignment, listnet distillation, fixed loss weights (three head launches at bs <= 64, the general kernels above)."""
        types = set(self.losses_types)
        return (not self.auto_weight and types <= {'matching', 'alignment', 'distillation'}
                and getattr(self.matching_criterion, 'sim', None) is dot_sim
                and ('alignment' not in types and 'distillation' not in types or self.alignment_criterion.aggregation == 'MrSw')
                and ('distillation' not in types or self.distillation_loss.mode == 'listnet'))

    def _graph_heads_now(self, img_emb, cap_emb, img_emb_set, cap_emb_seq, log):
        """graphed=True applies to a training step of a shipped configuration: gradients wanted, nothing being captured already."""
        return (self.graphed and not self._graph_bypass and log and torch.is_grad_enabled() and img_emb.is_cuda
                and any(t.requires_grad for t in (img_emb, cap_emb, img_emb_set, cap_emb_seq))
                and 'regularizehidden' not in self.losses_types and self._fused_heads_ok(img_emb)
                and not torch.cuda.is_current_stream_capturing())

    def forward_loss_total(self, img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, reg_loss, epoch=0,
                           distill_epoch=2, log=True):
        """forward_loss + the weighted sum of forward (alad_model.py:371-428 + :442-453) -> (loss, loss_dict).
        For the shipped configurations the whole thing is ONE autograd node (no element-wise glue; three head launches at
        bs <= 64): the terms of `loss_dict` are then detached values for logging, `loss` carries the graph.
        Otherwise it is forward_loss followed by weighted_total."""
        if self.shard_group is not False:
            return self._sharded_loss_total(img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, epoch,
                                            distill_epoch, log)
        if self._graph_heads_now(img_emb, cap_emb, img_emb_set, cap_emb_seq, log):
            if self._graph_step is None:
                from .graphs import GraphedLossStep
                self._graph_step = GraphedLossStep(self, log='deferred')
            return self._graph_step(img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, reg_loss, epoch, distill_epoch)
        if not self._fused_heads_ok(img_emb):
            d = self.forward_loss(img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, reg_loss, log=log)
            return self.weighted_total(d, epoch, distill_epoch), d
        wants_matching = 'matching' in self.config['training']['loss-type']
        heads = [k for k in ('matching', 'alignment', 'distillation')
                 if (k in self.losses_types) and (k != 'matching' or wants_matching)]
        logged_heads = list(heads)
        if epoch < distill_epoch and len(heads) > 1 and 'distillation' in heads:       # :442-444
            heads.remove('distillation')
        total, terms, _, _ = ops.small_batch_loss_heads(
            img_emb, cap_emb, img_emb_set.permute(1, 0, 2), cap_emb_seq.permute(1, 0, 2), img_lengths, cap_lengths,
            self.matching_criterion.margin, self.matching_criterion.max_violation, logged_heads, 
            {k: (self.losses_weights[k] if k in heads else 0.0) for k in logged_heads})
        idx = {'matching': 0, 'alignment': 1, 'distillation': 2}
        names = {'matching': 'matching_loss', 'alignment': 'alignment_loss', 'distillation': 'distillation_loss'}
        self.pending_log = [(names[k], terms[idx[k]], img_emb.size(0)) for k in logged_heads]
        if log:
            self.flush_log()
        return total, {k: terms[idx[k]] for k in heads}

    def _sharded_loss_total(self, img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, epoch, distill_epoch,
                            log=True):
        """forward_loss_total on the global batch of all ranks (self.shard_group): the same terms, gating, weights and
        logger keys as the single-device path on the concatenated batch; logged with n = the GLOBAL batch size."""
        from . import distributed as AD
        import torch.distributed as dist
        if not self._fused_heads_ok(img_emb):
            raise NotImplementedError('aladin_amd: the sharded step covers the shipped configurations (dot measure, MrSw, '
                                      'listnet, fixed loss weights); got %r' % (self.config['training'],))
        wants_matching = 'matching' in self.config['training']['loss-type']
        logged_heads = [k for k in ('matching', 'alignment', 'distillation')
                        if (k in self.losses_types) and (k != 'matching' or wants_matching)]
        heads = list(logged_heads)
        if epoch < distill_epoch and len(heads) > 1 and 'distillation' in heads:       # :442-444
            heads.remove('distillation')
        mc = self.matching_criterion
        total, terms, _, _ = AD.sharded_loss_heads(
            img_emb, cap_emb, img_emb_set.permute(1, 0, 2), cap_emb_seq.permute(1, 0, 2), img_lengths, cap_lengths,
            mc.margin, mc.max_violation, logged_heads, {k: (self.losses_weights[k] if k in heads else 0.0) for k in logged_heads},
            group=self.shard_group)
        n = img_emb.size(0) * dist.get_world_size(self.shard_group)
        names = {'matching': 'matching_loss', 'alignment': 'alignment_loss', 'distillation': 'distillation_loss'}
        self.pending_log = [(names[k], terms[k], n) for k in logged_heads]
        if log:
            self.flush_log()
        return total, {k: terms[k] for k in heads}

    def forward(self, example_imgs, example_txts, epoch=0, distill_epoch=2):
        """reference alad_model.py:430-454."""
        self.Eiters += 1
        if self._logger is not None:
            self._logger.update('Eit', self.Eiters)
        img_emb_aggr, cap_emb_aggr, img_feats, cap_feats, img_lengths, cap_lengths, regul_loss = \
            self.forward_emb(example_imgs, example_txts)
        return self.forward_loss_total(img_emb_aggr, cap_emb_aggr, img_feats, cap_feats, img_lengths, cap_lengths, regul_loss,
                                       epoch, distill_epoch)
