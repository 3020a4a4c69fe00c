import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import synth
from zippy_amd import api
from zippy_amd._binding import Engine
import zlib
stream = torch.cuda.current_stream()
eng = Engine(api.LIB_PATH, stream=stream.cuda_stream)
eng.set_gzip_fname_len(0)
src = synth.gen_batch("mix", 1, 1 << 20)[0].tobytes()
for name, comp in (("own", None), ("zlib6", zlib.compress(src, 6))):
    if comp is None:
        comp = eng.compress(src, 1, api.dfGzip)
        fmt = api.dfGzip
    else:
        fmt = api.dfZlib
    d_c = torch.frombuffer(bytearray(comp + b"\0" * 64), dtype=torch.uint8).cuda()
    d_o = torch.empty(len(src) + 64, dtype=torch.uint8, device="cuda")
    plan = eng.plan_uncompress([0], [len(comp)], [0], [len(src)], fmt)
    plan.set_profiling(True)
    for _ in range(3):
        t = time.perf_counter()
        plan.run(d_c.data_ptr(), d_o.data_ptr())
        lens, sts = plan.results()
        dt = time.perf_counter() - t
    assert sts == [0] and d_o[:len(src)].cpu().numpy().tobytes() == src
    print(name, len(comp), "wall %.3f ms" % (dt * 1e3), {k: round(v, 3) for k, v in plan.kernel_times()})
