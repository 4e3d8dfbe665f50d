# Round-end profile set: rocprofv3 kernel stats of the bench commands, and FETCH_SIZE / WRITE_SIZE / SQ counter
# passes (one run each) of tools/pmc_probe.py with the e4m3 formats.  Run on the GPU box from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1d
mkdir -p $O
cd $R
# kernel stats of the default bench command and of one engine alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/e4 -o e4 --output-format csv -- python3 bench.py --no-cpu-baseline --also-batch 0 > $O/e4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/e1 -o e1 --output-format csv -- python3 bench.py --no-cpu-baseline --also-batch 0 --engines 1 --steps 3 --warmup 1 > $O/e1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f8 -o f8 --output-format csv -- python3 bench.py --no-cpu-baseline --also-batch 0 --engines 1 --steps 3 --warmup 1 --weights fp8 --kv fp8 > $O/f8.log 2>&1
# one engine at 128 chunks per pass (what the scheduler's coalescing runs), and beam search (beam 5, 16 recordings per pass)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/b128 -o b128 --output-format csv -- python3 bench.py --no-cpu-baseline --also-batch 0 --batch 128 --engines 1 --steps 2 --warmup 1 > $O/b128.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/beam -o beam --output-format csv -- python3 tools/beam_probe.py 5 16 1 2 > $O/beam.log 2>&1
# counters, fp8 formats, one pass each
export MW_WEIGHTS=fp8 MW_KV=fp8
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/p_fetch -o p --output-format csv -- python3 tools/pmc_probe.py > $O/p_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/p_write -o p --output-format csv -- python3 tools/pmc_probe.py > $O/p_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/p_sq -o p --output-format csv -- python3 tools/pmc_probe.py > $O/p_sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_kt -o p --output-format csv -- python3 tools/pmc_probe.py > $O/p_kt.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fp8.csv $O/p_fetch $O/p_write $O/p_sq
# keep the summaries only: the per-dispatch traces are hundreds of MB
mkdir -p $O/keep
for n in e4 e1 f8 b128 beam; do cp $(find $O/$n -name "*_kernel_stats.csv") $O/keep/kernel_stats_$n.csv; done
cp $(find $O/p_kt -name "*_kernel_stats.csv") $O/keep/kernel_stats_pmc_probe_fp8.csv
cp $O/pmc_fp8.csv $O/*.log $O/keep/
rm -rf $O/e4 $O/e1 $O/f8 $O/b128 $O/beam $O/p_fetch $O/p_write $O/p_sq $O/p_kt
ls $O/keep
echo PROF_DONE
