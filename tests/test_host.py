"""Host logic that needs no GPU: audio decoding, chunk scheduling, the multi-rank exchange (gloo,
world size 2), and that the C-ABI library loads and exports every declared symbol."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import manual_whisper_amd as mw
from manual_whisper_amd import audio, parallel, scheduler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wav_roundtrip_and_resample(tmp_path):
    t = np.arange(16000) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    p = tmp_path / "a.wav"
    audio.write_wav(str(p), x)
    y = mw.load_audio(str(p))
    assert y.dtype == np.float32 and y.shape == x.shape and np.abs(y - x).max() < 1e-4
    p8 = tmp_path / "b.wav"
    audio.write_wav(str(p8), x[::2], rate=8000)               # 8 kHz -> 16 kHz
    z = mw.load_audio(str(p8))
    assert abs(len(z) - 16000) <= 2 and np.abs(z[200:-200] - x[200:len(z) - 200]).max() < 0.05
    with pytest.raises((RuntimeError, ValueError)):
        q = tmp_path / "c.m4a"
        q.write_bytes(b"not audio at all")
        mw.load_audio(str(q))


def test_split_chunks_edges():
    c, s = scheduler.split_chunks(np.zeros(0, np.float32))
    assert c.shape == (0, 480000) and s == []
    c, s = scheduler.split_chunks(np.ones(480000, np.float32))
    assert c.shape == (1, 480000) and s == [(0.0, 30.0)]
    c, s = scheduler.split_chunks(np.ones(480001, np.float32))
    assert c.shape == (2, 480000) and c[1, 0] == 1 and (c[1, 1:] == 0).all() and s[1] == (30.0, 480001 / 16000)
    assert scheduler.batches_of(5, 2) == [(0, 2), (2, 4), (4, 5)] and scheduler.batches_of(0, 4) == []


def test_group_sizes():
    assert scheduler.group_sizes(0, 4) == []
    assert scheduler.group_sizes(3, 4) == [3] and scheduler.group_sizes(4, 4) == [4]
    assert scheduler.group_sizes(5, 4) == [3, 2] and scheduler.group_sizes(20, 4) == [4] * 5
    assert scheduler.group_sizes(9, 4) == [3, 3, 3] and scheduler.group_sizes(7, 1) == [1] * 7
    for n in range(1, 40):
        for g in (1, 2, 4, 8):
            sz = scheduler.group_sizes(n, g)
            assert sum(sz) == n and max(sz) <= g and max(sz) - min(sz) <= 1 and len(sz) == (n + g - 1) // g


class _StandInEngine:
    """Host-only stand-in for engine.Engine in GroupedPool: an "encoder pass" writes each chunk's first sample into
    the decode engine's slots, a "decode loop" returns them; the log records who did what, in which order."""
    def __init__(self, max_batch, log):
        self.max_batch, self.log, self.slots, self.rows = max_batch, log, {}, 0

    def fork(self, batch):
        return _StandInEngine(batch, self.log)

    def set_decode_options(self, opts):
        pass

    def set_option(self, key, value):
        self.log.append(("option", key, value))

    def encode_into(self, pcm, dst, slot0):
        assert pcm.shape[0] <= self.max_batch
        for r in range(pcm.shape[0]):
            dst.slots[slot0 + r] = int(pcm[r, 0])
        self.log.append(("encode", id(dst), slot0, pcm.shape[0]))

    def set_encoded(self, rows):
        self.rows = rows

    def decode_greedy(self, rows):
        assert rows == self.rows and rows <= self.max_batch and sorted(self.slots)[:rows] == list(range(rows))
        ids = np.array([[self.slots[r]] for r in range(rows)], np.int32)
        self.log.append(("decode", id(self), rows))
        self.slots = {}
        return ids, np.ones(rows, np.int32), None

    def close(self):
        pass


def test_grouped_pool_order_and_group_split(monkeypatch):
    """GroupedPool on stand-in engines: every batch comes back in its place whatever the split into decode groups,
    a decode engine's cache is refilled only after its previous group has been decoded, and the optional trace lists
    every encoder pass and decode loop."""
    monkeypatch.setenv("MW_POOL_TRACE", "1")
    log = []
    pool = scheduler.GroupedPool(_StandInEngine(4, log), batch=4, group=3, n_encoders=2, n_decoders=2)
    try:
        batches = [np.full((4 if i % 3 else 2, 8), 100 + i, np.float32) for i in range(11)]
        for sizes in (None, [1, 3, 2, 3, 2], [3, 3, 3, 2]):
            del log[:]
            got = pool.map(batches, sizes=sizes)
            assert [int(ids[0, 0]) for ids, _ in got] == [100 + i for i in range(11)]
            assert all(ids.shape[0] == b.shape[0] for (ids, _), b in zip(got, batches))
            want = sizes or scheduler.group_sizes(11, 3)
            assert sorted(k for k, *_ in pool.trace) == ["decode"] * len(want) + ["encode"] * 11
            # per decode engine: encoder passes, a decode loop, encoder passes, a decode loop: never a pass into a cache
            # still to be decoded; the two encoders finish in any order, and the passes of a group tile its slots
            for dec in pool.decoders:
                pieces = []
                for e in log:
                    if e[0] == "encode" and e[1] == id(dec):
                        pieces.append((e[2], e[3]))
                    elif e[0] == "decode" and e[1] == id(dec):
                        at = 0
                        for slot0, rows in sorted(pieces):
                            assert slot0 == at
                            at += rows
                        assert e[2] == at
                        pieces = []
                assert pieces == []
        for bad in ([3, 3, 3, 3], [4, 4, 3], [11], [3, 3, 3, 2, 0]):
            with pytest.raises(ValueError):
                pool.map(batches, sizes=bad)
        assert pool.map([]) == []
    finally:
        pool.close()


def test_model_pool_stays_within_the_sequence_cap_when_batch_size_changes():
    """Model._pool_for over stand-in engines: transcribe(batch_size=16), then 64, then 128, then 16 again.  A decode
    engine never holds more than max_group_sequences sequences (batch * group), a larger batch shrinks the group, and
    a call the pool at hand can serve does not rebuild it."""
    from manual_whisper_amd.model import Model
    log, built = [], []

    class _M(Model):
        def _make_grouped_pool(self, batch, group, n_encoders, n_decoders):
            built.append((batch, group, n_encoders, n_decoders))
            return scheduler.GroupedPool(_StandInEngine(16, log), batch, group, n_encoders, n_decoders)

    m = _M(_StandInEngine(16, log), mw.get_dims("micro"), "zh", arrangement="grouped", synthetic=True)
    for bs, n_chunks in ((16, 320), (64, 640), (128, 640), (16, 320), (16, 40), (128, 128)):
        pool = m._pool_for(n_chunks, bs)
        batch, group = m._pool_key[1:3]
        assert batch >= bs and batch * group <= m.max_group_sequences and len(pool.decoders) <= 3
        assert pool.batch == batch and pool.group == group
    assert built == [(16, 7, 2, 3), (64, 2, 2, 3), (128, 1, 2, 3)], built
    m._close_pool()


def test_hardware_queues_are_raised_before_torch_or_hip_load():
    """The command line path (`import transcribe`, as `python transcribe.py x` does before main()) puts
    GPU_MAX_HW_QUEUES into the environment before torch is imported or a HIP runtime is mapped: 8 by default whatever
    the environment said, MW_HW_QUEUES to override, never above 32."""
    probe = ("import os, sys\n"
             f"sys.path.insert(0, {ROOT!r})\n"
             "import transcribe\n"
             "from manual_whisper_amd import _lib\n"
             "assert 'torch' not in sys.modules and not _lib._hip_runtime_mapped()\n"
             "print('queues', os.environ['GPU_MAX_HW_QUEUES'], _lib.configure_hw_queues())\n")
    for env_in, want in (({"GPU_MAX_HW_QUEUES": "4"}, "8"), ({"MW_HW_QUEUES": "12"}, "12"), ({"MW_HW_QUEUES": "64"}, "32"),
                         ({"MW_HW_QUEUES": "junk"}, "8")):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "MW_HW_QUEUES")}
        env.update(env_in)
        r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0 and f"queues {want} {want}" in r.stdout, r.stdout + r.stderr


def test_pools_warn_when_streams_outnumber_hardware_queues(monkeypatch):
    from manual_whisper_amd import _lib
    monkeypatch.setitem(_lib._hw_queues, "effective", 4)
    monkeypatch.setitem(_lib._hw_queues, "late", True)
    with pytest.warns(RuntimeWarning, match="hardware queues"):
        scheduler.GroupedPool(_StandInEngine(4, []), batch=4, group=2, n_encoders=2, n_decoders=3).close()
    monkeypatch.setitem(_lib._hw_queues, "effective", 8)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        scheduler.GroupedPool(_StandInEngine(4, []), batch=4, group=2, n_encoders=2, n_decoders=3).close()


def test_shard_and_pack():
    for n in (0, 1, 7, 120):
        for w in (1, 2, 8):
            r = [parallel.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    ids = [np.array([3, 4, 5]), np.array([], np.int32), np.array([9])]
    rows = parallel.pack_rows(ids, 4, 5)
    assert rows.shape == (5, 5) and rows[3, 0] == -1
    back = parallel.unpack_rows(rows)
    assert len(back) == 3 and all((a == b).all() for a, b in zip(back, ids))


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch.distributed as dist
import manual_whisper_amd
from manual_whisper_amd import parallel
dist.init_process_group("gloo")
n = int(sys.argv[1])
chunks = np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 8), np.float32)
def fake(c):      # ids derived from the chunk's content; ragged lengths incl. empty
    return [np.arange(int(v[0]) %% 4, dtype=np.int32) + int(v[0]) * 10 for v in c]
out = parallel.transcribe_sharded(fake, chunks, max_new=5)
want = fake(chunks)
assert len(out) == n and all((a == b).all() for a, b in zip(out, want)), (out, want)
# the same with every rank cutting only its own share, the window list coming from rank 0
wins = parallel.broadcast_from_rank0(list(range(n)) if dist.get_rank() == 0 else None)
assert wins == list(range(n))
taken = []
def take(a, b):
    taken.append((a, b)); return chunks[a:b]
out = parallel.transcribe_sharded(fake, take, max_new=5, n_items=len(wins))
assert len(out) == n and all((a == b).all() for a, b in zip(out, want)), (out, want)
assert taken == ([parallel.shard_range(n, dist.get_rank(), 2)] if parallel.shard_range(n, dist.get_rank(), 2)[1] > parallel.shard_range(n, dist.get_rank(), 2)[0] else [])
# forced alignment sharded the same way: each rank "aligns" its block, one gather, order kept on every rank
segs = [{"text": "s%%d" %% i, "start": float(i), "end": i + 0.5} for i in range(n)]
seen = []
def fake_align(block):
    seen.append([s_["text"] for s_ in block])
    words = [{"word": s_["text"] + w, "start": s_["start"], "end": s_["end"], "score": 1.0} for s_ in block for w in ("a", "b")]
    return {"segments": [dict(s_, words=[w for w in words if w["word"].startswith(s_["text"] + "a") or w["word"].startswith(s_["text"] + "b")]) for s_ in block],
            "word_segments": words}
got = parallel.align_sharded(fake_align, segs)
a_, b_ = parallel.shard_range(n, dist.get_rank(), 2)
assert seen == ([[s_["text"] for s_ in segs[a_:b_]]] if b_ > a_ else []), seen
assert [s_["text"] for s_ in got["segments"]] == [s_["text"] for s_ in segs]
assert [w["word"] for w in got["word_segments"]] == [s_["text"] + w for s_ in segs for w in ("a", "b")]
print("rank", dist.get_rank(), "ok")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_chunks", [0, 1, 5, 8])
def test_sharded_gather_world2_gloo(tmp_path, n_chunks):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29500 + n_chunks + os.getpid() % 200),
                        str(script), str(n_chunks)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_library_exports_every_declared_symbol():
    from manual_whisper_amd import _lib
    header = open(os.path.join(ROOT, "include", "mwhip.h")).read()
    declared = set(re.findall(r"\b(mw_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_load_model_rejects_what_it_cannot_do():
    from manual_whisper_amd.engine import EngineError
    with pytest.raises(EngineError, match="GPU only"):
        mw.load_model("synthetic:micro", "cpu")
    with pytest.raises(EngineError, match="compute_type"):
        mw.load_model("synthetic:micro", "cuda", compute_type="int8")
    with pytest.raises(FileNotFoundError, match="Nothing is downloaded"):
        mw.load_model("large-v3", "cuda")


def test_bench_starts_its_own_ranks_before_touching_torch(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must hand over to torch.distributed.run from a process
    that has imported neither torch nor the HIP library."""
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, os\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "os.environ.pop('WORLD_SIZE', None)\n"
        "import subprocess, bench\n"
        "seen = {}\n"
        "def fake_call(cmd, **kw):\n"
        "    seen['cmd'] = cmd\n"
        "    seen['torch'] = 'torch' in sys.modules\n"
        "    seen['lib'] = 'manual_whisper_amd._lib' in sys.modules and sys.modules['manual_whisper_amd._lib']._lib is not None\n"
        "    return 7\n"
        "subprocess.call = fake_call\n"
        "try:\n"
        "    bench.main(['--gpus', '2', '--steps', '3', '--warmup', '1'])\n"
        "except SystemExit as e:\n"
        "    seen['rc'] = e.code\n"
        "cmd = seen['cmd']\n"
        "assert seen['rc'] == 7 and not seen['torch'] and not seen['lib'], seen\n"
        "assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=2' in cmd, cmd\n"
        "assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '2', '--steps', '3', '--warmup', '1'], cmd\n"
        "print('launcher ok')\n")
    r = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "launcher ok" in r.stdout, r.stdout + r.stderr


def test_bench_with_two_gpus_fails_only_for_want_of_a_gpu():
    """On a box without GPUs the self-launched ranks must get as far as asking for their device."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in out
    assert "No HIP GPUs" in out or "no such HIP device" in out or "Found no NVIDIA driver" in out or "CUDA" in out or "HIP" in out, out[-2000:]
