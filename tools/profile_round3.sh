# Profile set behind profiles/round3_*.csv (run on the GPU box from the repo root): rocprofv3 kernel stats of the bench commands
# (the default arrangement, one engine at 16 and at 112 sequences), of the 75-id prompt arrangement and of the forced aligner, and
# FETCH_SIZE / WRITE_SIZE passes (one run each) of tools/pmc_probe.py at 16 and at 112 sequences.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O/keep
cd $R
st() {   # st <tag> <program and its arguments...>
  tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/$tag -o $tag --output-format csv -- "$@" > $O/keep/$tag.log 2>&1
  cp $(find $O/$tag -name "*_kernel_stats.csv") $O/keep/kernel_stats_$tag.csv
  rm -rf $O/$tag
  echo "stats $tag done"
}
st grouped python3 bench.py --only-main --steps 20 --warmup 5
st e1 python3 bench.py --only-main --group 1 --engines 1 --steps 3 --warmup 1
st batch112_e1 python3 bench.py --only-main --group 1 --engines 1 --batch 112 --steps 2 --warmup 1
st prompt75 python3 tools/prompt_probe.py 75 14
st align python3 tools/align_bench.py
pmc() {  # pmc <tag> : FETCH_SIZE and WRITE_SIZE of tools/pmc_probe.py under the current MW_* environment
  tag=$1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf_$tag -o p --output-format csv -- python3 tools/pmc_probe.py > $O/keep/pmc_fetch_$tag.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw_$tag -o p --output-format csv -- python3 tools/pmc_probe.py > $O/keep/pmc_write_$tag.log 2>&1
  python3 tools/pmc_summary.py $O/keep/pmc_counters_$tag.csv $O/pf_$tag $O/pw_$tag
  rm -rf $O/pf_$tag $O/pw_$tag
  echo "pmc $tag done"
}
pmc f16
MW_B=112 pmc f16_b112
MW_B=112 MW_SHARED=1 pmc f16_b112_shared
st beam5_e3 python3 tools/beam_probe.py 5 48 3 2
ls $O/keep
echo PROF_DONE
