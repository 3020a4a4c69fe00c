import sys, zlib
sys.path.insert(0, '/root/repo')
import torch
torch.cuda.init()
import synth
from zippy_amd import api
import os
from zippy_amd._binding import Engine
eng = Engine(os.environ.get('ZH_LIB', api.LIB_PATH))
names = ["paper-100k.pdf.gz"]
for n in names:
    b = synth.fixture(n)
    want = zlib.decompress(b, 47)
    raw = b[10:-8]
    outs, sts = eng.uncompress_batch([raw], 3)
    o, st = outs[0], sts[0]
    if st != 0:
        print(n, "status", st); continue
    diffs = [i for i in range(min(len(o), len(want))) if o[i] != want[i]]
    print(n, "len", len(o), len(want), "ndiff", len(diffs), "first", diffs[:20])
    if diffs:
        k = diffs[0]
        print(" got ", o[k-8:k+24]); print(" want", want[k-8:k+24])
        # runs of diffs
        runs=[]; s=diffs[0]; p=s
        for d in diffs[1:]:
            if d != p+1: runs.append((s,p-s+1)); s=d
            p=d
        runs.append((s,p-s+1)); print(" runs", runs[:20])
