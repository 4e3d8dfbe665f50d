# Where the encoder attention's wave cycles go (run on the GPU box from the repo root): two counter passes over
# tools/attn_bench at the large-v3 batch-16 shape, summarised per kernel.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attnpmc
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 60 tools/attn_bench 16 20 1500 > $O/plain.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $O/p1 -o p --output-format csv -- tools/attn_bench 16 20 1500 > $O/run1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SALU -d $O/p2 -o p --output-format csv -- tools/attn_bench 16 20 1500 > $O/run2.log 2>&1
python3 tools/pmc_summary.py $O/attn.csv $O/p1 $O/p2
rm -rf $O/p1 $O/p2
cat $O/plain.log $O/attn.csv
