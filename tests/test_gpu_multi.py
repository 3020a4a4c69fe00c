"""Multi-GPU plumbing on the one GPU a test box has (-m gpu): zippy_amd/sharding.py over RCCL
("nccl") with world size 1, bench.py under torch.distributed.run with its transfer leg, and
bench.py refusing to report an N-GPU number from fewer than N devices.  The 2-rank logic of the
same functions runs on CPU in tests/test_sharding_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_gpu_sharding_over_nccl_world1():
    code = r"""
import sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from zippy_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
n, size = 7, 5000
batch = torch.randint(0, 256, (n * size,), dtype=torch.uint8, device="cuda")
mine = sharding.scatter_fixed(batch, n, size, root=0, device="cuda")
assert torch.equal(mine, batch)
lens = torch.tensor([100, 0, 4999, 1, 2500, 17, 5000], dtype=torch.int64, device="cuda")
packed = torch.cat([batch[i * size:i * size + int(l)] for i, l in enumerate(lens.tolist())])
data, all_lens = sharding.gather_variable(packed, lens, root=0)
assert torch.equal(data, packed) and all_lens.tolist() == lens.tolist()
back, my_lens = sharding.scatter_variable(data, all_lens, n, root=0, device="cuda")
assert torch.equal(back, packed) and my_lens.tolist() == lens.tolist()
home = sharding.gather_fixed(mine, n, size, root=0)
assert torch.equal(home, batch)
# device-resident plans: slots -> streams back to back (zh_plan_pack), over the group, -> slots (zh_plan_unpack)
from zippy_amd import api
from zippy_amd._binding import Engine
eng = Engine(api.LIB_PATH, device=0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_gzip_fname_len(0)
text = (b"whole buffers are all that ever travels " * 200)[:size]
src = torch.frombuffer(bytearray(text * n), dtype=torch.uint8).cuda()
cap = size + size // 8 + 2048
slot = (cap + 255) & ~255
offs_in, offs_slot = [i * size for i in range(n)], [i * slot for i in range(n)]
slots = torch.zeros(n * slot, dtype=torch.uint8, device="cuda")
cplan = eng.plan_compress(offs_in, [size] * n, offs_slot, [cap] * n, 1, api.dfGzip)
cplan.run(src.data_ptr(), slots.data_ptr())
packed, plens = sharding.pack_plan(cplan, slots, n)
clens, csts = cplan.results()
assert all(x == 0 for x in csts) and plens.tolist() == clens and packed.numel() == sum(clens)
for i in range(n):
    o = sum(clens[:i])
    assert torch.equal(packed[o:o + clens[i]], slots[i * slot:i * slot + clens[i]])
data, all_lens = sharding.gather_variable(packed, plens, root=0)
mine_c, mine_lens = sharding.scatter_variable(data, all_lens, n, root=0, device="cuda")
slots2 = torch.full((n * slot,), 0x5a, dtype=torch.uint8, device="cuda")
back = torch.zeros(n * size, dtype=torch.uint8, device="cuda")
uplan = eng.plan_uncompress(offs_slot, [cap] * n, offs_in, [size] * n, api.dfGzip)
keep = sharding.unpack_into_plan(uplan, mine_c, mine_lens, slots2)
uplan.run(slots2.data_ptr(), back.data_ptr())
ulens, usts = uplan.results()
assert all(x == 0 for x in usts) and ulens == [size] * n and torch.equal(back, src)
dist.barrier()
dist.destroy_process_group()
print("sharding nccl ok")
""" % (ROOT, _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharding nccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_gpu_bench_refuses_more_gpus_than_present():
    import torch
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--buffers", "64",
                        "--steps", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "refusing" in (r.stdout + r.stderr)
    assert '"n_gpus"' not in r.stdout  # no result line at all


def test_gpu_bench_under_torchrun_with_transfer_leg():
    """One rank under torch.distributed.run: the RCCL group, strong sharding and the
    scatter/gather leg all run (with one rank they move nothing between devices)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--buffers", "96", "--size", "262144", "--steps", "1", "--warmup", "1",
           "--scaling", "strong", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["scaling"] == "strong"
    assert out["config"]["buffers_total"] == 96
    assert out["transfer"]["scatter_ms"] >= 0 and out["value_incl_transfer"] > 0
    assert out["roofline"]["frac"] > 0 and "compress" in out["roofline_passes"]
