"""aladin_amd -- ALADIN's alignment-scoring / matching-retrieval hot path on MI355X (gfx950).

Drop-in names (reference mesnico/ALADIN):
    aladin_amd.loss         AlignmentContrastiveLoss, ContrastiveLoss, DistillationLoss   (alad/loss.py)
    aladin_amd.alad_model   ALADModel.forward / forward_emb / forward_loss                (alad/alad_model.py)
    aladin_amd.encoder      JointTextImageTransformerEncoder: backbone hand-off + matching head  (alad/alad_model.py:29-247)
    aladin_amd.evaluation   compute_sim_matrix, compute_recall, recall, i2t, t2i,         (alad/evaluation.py,
                            encode_data, encode_data_packed                               alad/recall_auxiliary.py)
    aladin_amd.store        PackedSetStore: 16-bit length-packed device-resident evaluation store
    aladin_amd.distributed  caption-block sharding of the alignment loss over the GPUs of a node (RCCL)
Everything numeric runs in the HIP kernels of aladin_amd/csrc behind include/aladin_hip.h.
Importing the package does not need a GPU; calling any scoring function without the built
extension or without an AMD GPU raises.
"""
__version__ = '0.1.0'

from . import synth  # noqa: F401


def __getattr__(name):
    import importlib
    if name in ('ops', 'loss', 'alad_model', 'encoder', 'evaluation', 'distributed', 'store', '_lib'):
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
