#!/usr/bin/env python
"""Window-attention timings on chosen (H, W, C, heads) geometries (development aid): separates the cost of edge windows
(padding / cyclic-shift wrap-around, the general addressing path) from interior ones.
usage: python tools/attn_shapes.py H,W,C,nH [H,W,C,nH ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench as m  # noqa: E402

if __name__ == "__main__":
    m.STAGES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    m.attn()
