#!/bin/bash
# kernel time of the packed forced-alignment pass by (kernel, grid size): which launches of a pack cost what
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/alignprof2
rm -rf $O && mkdir -p $O
cd $R
MW_ALIGN_PACK=${MW_ALIGN_PACK:-240} MW_ALIGN_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 tools/align_bench.py > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1][:40] if "mw" in name else name[:40]
    key = (short, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", ""))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
with open("$O/by_shape.txt", "w") as out:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        line = f"{k[0]:42s} grid {k[1]:>9s} wg {k[2]:>4s} lds {k[3]:>7s}: {v[0]:5d} calls {v[1] / 1e3:8.2f} ms total {v[1] / v[0]:9.1f} us each {100 * v[1] / tot:5.1f} %"
        print(line); out.write(line + "\n")
    print("total", tot / 1e3, "ms")
PY
rm -rf $O/t
