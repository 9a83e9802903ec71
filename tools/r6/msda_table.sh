# MSDA batch table on the kernel clock: rocprofv3 --kernel-trace --stats of tools/msda_batch_table.py is not needed for the table itself (event clock over a
# backlogged queue); this script just runs it and keeps the output.  usage (GPU box): bash tools/r6/msda_table.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python3 tools/msda_batch_table.py gpurun_out/r6/r6_msda_batch_table.txt 2>&1 | grep -v amdgpu.ids | tail -12
