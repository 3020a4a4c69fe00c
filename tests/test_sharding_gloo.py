"""N > 1 path on CPU: 2, 8 and 16 gloo ranks shard a batch, each compresses its shard (kernel
sources under the emulator -- test infrastructure), the compressed buffers come back
to rank 0 in order and equal the oracle's output; then the reverse for uncompress."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_buffers, buf_bytes, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import emu
    import oracle
    import synth
    from zippy_amd import sharding
    eng = emu.engine()
    batch = None
    if rank == 0:
        batch = torch.from_numpy(synth.gen_batch("mix", n_buffers, buf_bytes).reshape(-1).copy())
    mine = sharding.scatter_fixed(batch, n_buffers, buf_bytes, root=0)
    lo, hi = sharding.shard_range(n_buffers, rank, world)
    assert mine.numel() == (hi - lo) * buf_bytes
    raw = mine.numpy().tobytes()
    bufs = [raw[i * buf_bytes:(i + 1) * buf_bytes] for i in range(hi - lo)]
    outs, sts = eng.compress_batch(bufs, 1, oracle.dfGzip) if bufs else ([], [])
    assert all(s == 0 for s in sts)
    local = torch.from_numpy(np.frombuffer(b"".join(outs) or b"\0", dtype=np.uint8).copy())
    lens = torch.tensor([len(o) for o in outs], dtype=torch.int64)
    data, all_lens = sharding.gather_variable(local, lens, root=0)
    ok = True
    if rank == 0:
        host = batch.numpy().tobytes()
        blob = data.numpy().tobytes()
        off = 0
        comp = []
        for i, ln in enumerate(all_lens.tolist()):
            piece = blob[off:off + ln]
            off += ln
            comp.append(piece)
            ok &= piece == oracle.compress(host[i * buf_bytes:(i + 1) * buf_bytes], 1, oracle.dfGzip, fname_len=0)
        ok &= len(comp) == n_buffers
    # and back: every rank uncompresses the shard it produced, rank 0 reassembles the batch
    back, sts = eng.uncompress_batch(outs) if outs else ([], [])
    assert all(s == 0 for s in sts)
    local = torch.from_numpy(np.frombuffer(b"".join(back) or b"\0", dtype=np.uint8).copy())
    lens = torch.tensor([len(o) for o in back], dtype=torch.int64)
    data2, all_lens2 = sharding.gather_variable(local, lens, root=0)
    if rank == 0:
        ok &= data2.numpy().tobytes() == batch.numpy().tobytes()
        ok &= all_lens2.tolist() == [buf_bytes] * n_buffers
    # the uncompress direction of a batch that lives on rank 0: compressed streams go out with
    # scatter_variable, results come home with gather_fixed
    mine_c, mine_lens = sharding.scatter_variable(data, all_lens if rank == 0 else None, n_buffers, root=0)
    assert mine_lens.tolist() == [len(o) for o in outs]
    assert mine_c.numpy().tobytes() == b"".join(outs)
    local = torch.from_numpy(np.frombuffer(b"".join(back) or b"\0", dtype=np.uint8).copy())
    whole = sharding.gather_fixed(local, n_buffers, buf_bytes, root=0)
    if rank == 0:
        ok &= whole.numpy().tobytes() == batch.numpy().tobytes()
    # the same trip with device-resident plans (here: host memory under the emulator): the shard compressed into
    # worst-case slots, the streams packed back to back by zh_plan_pack (no per-buffer copy), home with
    # gather_variable, out again with scatter_variable, into the uncompress plan's slots by zh_plan_unpack
    k = hi - lo
    if k:
        cap = buf_bytes + buf_bytes // 8 + 2048
        slot = (cap + 255) & ~255
        offs_in, offs_slot = [i * buf_bytes for i in range(k)], [i * slot for i in range(k)]
        d_src = mine.clone()
        d_slots = torch.zeros(k * slot, dtype=torch.uint8)
        cplan = eng.plan_compress(offs_in, [buf_bytes] * k, offs_slot, [cap] * k, 1, oracle.dfGzip)
        cplan.run(d_src.data_ptr(), d_slots.data_ptr())
        packed, plens = sharding.pack_plan(cplan, d_slots, k)
        assert plens.tolist() == [len(o) for o in outs] and packed.numpy().tobytes() == b"".join(outs)
    else:
        packed, plens = torch.zeros(1, dtype=torch.uint8)[:0], torch.zeros(0, dtype=torch.int64)
    data3, all_lens3 = sharding.gather_variable(packed if k else torch.zeros(1, dtype=torch.uint8), plens, root=0)
    if rank == 0:
        ok &= data3.numpy().tobytes() == data.numpy().tobytes() and all_lens3.tolist() == all_lens.tolist()
    mine_c3, mine_lens3 = sharding.scatter_variable(data3, all_lens3 if rank == 0 else None, n_buffers, root=0)
    if k:
        d_slots2 = torch.full((k * slot,), 0x5a, dtype=torch.uint8)
        d_back = torch.zeros(k * buf_bytes, dtype=torch.uint8)
        uplan = eng.plan_uncompress(offs_slot, [cap] * k, offs_in, [buf_bytes] * k, oracle.dfGzip)
        keep = sharding.unpack_into_plan(uplan, mine_c3, mine_lens3, d_slots2)
        uplan.run(d_slots2.data_ptr(), d_back.data_ptr())
        ulens, usts = uplan.results()
        assert all(x == 0 for x in usts) and ulens == [buf_bytes] * k
        assert torch.equal(d_back, mine)
        local3 = d_back
    else:
        local3 = torch.zeros(1, dtype=torch.uint8)
    whole3 = sharding.gather_fixed(local3, n_buffers, buf_bytes, root=0)
    if rank == 0:
        ok &= whole3.numpy().tobytes() == batch.numpy().tobytes()
        with open(result_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


# (world, buffers): two ranks with an uneven and an even split; BASELINE's eight ranks with 13 buffers (shards of two and of
# one); sixteen ranks with 13 buffers: three ranks own NOTHING and still take part in every collective
@pytest.mark.parametrize("world,n_buffers", [(2, 5), (2, 2), (8, 13), (16, 13)])
def test_ranks_shard_compress_gather(tmp_path, world, n_buffers):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    build_emu.build()  # once, before the ranks race for it
    import oracle
    oracle.build()
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), n_buffers, 20000 if world == 2 else 9000, result), nprocs=world, join=True)
    assert open(result).read() == "ok"


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from zippy_amd import sharding
    for n in (0, 1, 7, 8, 4096):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
