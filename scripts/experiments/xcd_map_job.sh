cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export IHMR_HIP_LIBRARY=$PWD/build/xcdmap.so
timeout 600 python -m pytest tests/test_gpu_encoder.py -q -m gpu -x 2>&1 | tail -2
unset IHMR_HIP_LIBRARY
MODE=baseline REPS=2 bash scripts/ab.sh build/head2.so build/xcdmap.so 2>&1 | grep -v amdgpu
for lib in head2 xcdmap; do
  export IHMR_HIP_LIBRARY=$PWD/build/$lib.so
  rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --config baseline --no-cpu-baseline > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --config baseline --no-cpu-baseline > /dev/null 2>&1
  python3 scripts/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/xcd_${lib}_traffic.csv > /dev/null
  echo "== $lib"; grep "conv_" gpurun_out/xcd_${lib}_traffic.csv | head -30
  rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
done
