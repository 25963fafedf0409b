"""Stand-in for bench.py's rank body, for tests/test_launch_cpu.py: every rank joins a gloo group on the rendezvous the
launcher prepared, the ranks count themselves with an all-reduce, rank 0 prints noise, then ONE JSON line, then
(like RCCL's banner) every rank prints more noise unless told to flush properly.  --fail-rank K makes rank K exit 3."""
import argparse
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument('--gpus', type=int, default=1)
ap.add_argument('--fail-rank', type=int, default=-1)
ap.add_argument('--no-result', action='store_true')
ap.add_argument('--hang', action='store_true')
ap.add_argument('--fail-after-result', action='store_true')
ap.add_argument('--bind-failure-once', default='')
args = ap.parse_args()

import torch
import torch.distributed as dist

rank, world, local = (int(os.environ[k]) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'))
if args.bind_failure_once and not os.path.exists(args.bind_failure_once):
    # first attempt only: what torch.distributed.run prints when another job took the rendezvous port
    if rank == 0:
        open(args.bind_failure_once, 'w').write('x')
        print('RuntimeError: The server socket has failed to listen on any local network address. port: 1: Address already in use', file=sys.stderr)
    sys.exit(1)
if rank == args.fail_rank:
    print('rank %d: failing on purpose' % rank, file=sys.stderr)
    sys.exit(3)
if args.hang:
    import time
    time.sleep(600)
dist.init_process_group('gloo')
t = torch.zeros(world, dtype=torch.int64)
t[rank] = 1 + local
dist.all_reduce(t)
if rank == 0:
    print('banner: not the result line')
    if not args.no_result:
        print('{"looks": "like json but is followed by the real line"}')
        print(json.dumps({'n_gpus': world, 'asked': args.gpus, 'local_ranks_plus_1': t.tolist(),
                          'master': [os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')],
                          'self_launched': os.environ.get('ALADIN_SELF_LAUNCHED')}), flush=True)
dist.barrier()
if args.fail_after_result and rank == world - 1:
    print('rank %d: failing after the result line' % rank, file=sys.stderr)
    sys.exit(5)
print('rank %d late noise' % rank, flush=True)
dist.destroy_process_group()
