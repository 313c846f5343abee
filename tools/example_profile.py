#!/usr/bin/env python
"""cProfile of the timed part of examples/selfplay_train.py (host side of the whole training loop): top functions by
internal and by cumulative time.  Usage (GPU box): python tools/example_profile.py [example args]"""
import cProfile
import io
import os
import pstats
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "examples"))
sys.argv = ["selfplay_train.py"] + (sys.argv[1:] or ["--iters", "40"])
import selfplay_train as m  # noqa: E402

pr = cProfile.Profile()
orig, seen = torch.cuda.synchronize, [0]


def sync(*a, **k):          # the example synchronises once when its timed part starts: profile from there
    orig(*a, **k)
    seen[0] += 1
    if seen[0] == 1:
        pr.enable()


torch.cuda.synchronize = sync
m.main()
pr.disable()
for key, n in (("tottime", 30), ("cumulative", 60)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue())
