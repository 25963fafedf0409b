#!/usr/bin/env python3
"""sha256 over what the kernels are built from: aladin_amd/csrc/*.hip, *.hpp, the Makefile with its flags, include/aladin_hip.h (sorted
by name; first 16 hex digits).  Three users, one definition:
  * the Makefile writes it next to each library it links (lib*.so.srchash): the sources THAT binary was built from;
  * bench.py / tests compare that stamp with the sources of the tree (a library left over from an experiment is reported, not timed
    as if it were the committed code);
  * committed PMC summaries (profiles/*_pmc.json) carry it, and bench.py refuses a summary collected on other sources.
No third-party imports: the Makefile calls this."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    csrc = os.path.join(root, 'aladin_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.hpp')) + [os.path.join(csrc, 'Makefile')]) + \
        [os.path.join(root, 'include', 'aladin_hip.h')]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def library_stamp(lib_path):
    """the source hash recorded when `lib_path` was linked, or None (a library built before the stamp existed, or by hand)"""
    try:
        return open(lib_path + '.srchash').read().strip() or None
    except OSError:
        return None


if __name__ == '__main__':
    sys.stdout.write(csrc_hash() + '\n')
