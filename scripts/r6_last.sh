python3 -m pytest tests -q -m gpu -rP > gpurun_out/r6_gputests_full2.log 2>&1; tail -3 gpurun_out/r6_gputests_full2.log
bash scripts/r6_final.sh r6_v2
