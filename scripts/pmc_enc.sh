#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pe
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pe -o e -- python3 bench.py --config baseline --no-cpu-baseline > gpurun_out/pe.log 2>&1
python3 scripts/pmc_sq_summary.py gpurun_out/pe gpurun_out/pe_sq.csv | grep -E "kernel|conv_igemm" | cut -c1-400
rm -rf gpurun_out/pe
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pe -o e -- python3 bench.py --config baseline --no-cpu-baseline > gpurun_out/pe.log 2>&1
python3 scripts/pmc_sq_summary.py gpurun_out/pe gpurun_out/pe_sq2.csv | grep -E "kernel|conv_igemm" | cut -c1-400
rm -rf gpurun_out/pe
