"""Development aid: host-clock timeline of scheduler.GroupedPool over a finite job (the benchmark's timed region is
one map() of `steps` batches from an idle GPU to a drained one).

    MW_POOL_TRACE=1 python tools/pool_trace.py [steps] [group] [decoders] [encoders] [sizes]

`sizes`: comma-separated group sizes (batches per decode group, in order) instead of the even split."""
import os
import sys
import time

os.environ.setdefault("MW_POOL_TRACE", "1")
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("MW_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import manual_whisper_amd as mw  # noqa: E402
from manual_whisper_amd import synth  # noqa: E402
from manual_whisper_amd.decode_options import synthetic_options  # noqa: E402
from manual_whisper_amd.engine import Engine  # noqa: E402
from manual_whisper_amd.scheduler import GroupedPool  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    group = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    n_dec = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    n_enc = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    sizes = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else None
    dims = mw.get_dims("large-v3")
    dev = torch.device("cuda:0")
    weights = synth.make_weights_torch(dims, seed=1234, device=dev)
    eng = Engine(dims, device=0, max_batch=16)
    eng.load_weights(weights)
    opts = synthetic_options(dims, max_new=128, suppress_eot=True)
    eng.set_decode_options(opts)
    pool = GroupedPool(eng, 16, group, n_enc, n_dec)
    pool.set_decode_options(opts)
    del weights
    pcms = [synth.make_pcm_torch(16, seed=i, device=dev) for i in range(4)]
    torch.cuda.synchronize()
    pool.map([pcms[0]] * (group * n_dec))
    pool.map([pcms[i % 4] for i in range(5)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if sizes:
        pool.map([pcms[i % 4] for i in range(steps)], sizes=sizes)
    else:
        pool.map([pcms[i % 4] for i in range(steps)])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"steps {steps} group {group} decoders {n_dec} encoders {n_enc} sizes {sizes}: {1e3 * (t1 - t0):.1f} ms, {1e3 * (t1 - t0) / steps:.2f} ms per step")
    for kind, i, a, b in sorted(pool.trace, key=lambda r: r[2]):
        print(f"  {kind:6s} {i:3d}  {1e3 * (a - t0):8.1f} -> {1e3 * (b - t0):8.1f}  ({1e3 * (b - a):7.1f} ms)")
    pool.close()


if __name__ == "__main__":
    main()
