export IHMR_HIP_LIBRARY=$PWD/build/ab/tune.so
scripts/prof_encoder_layers.sh t_def > /dev/null
IHMR_CONV_FORCE="3 1" IHMR_CONV_SK="0 0 512" scripts/prof_encoder_layers.sh t_64x64 > /dev/null
paste <(cut -c1-14 gpurun_out/t_def_encoder_layers.txt) <(cut -c4-14 gpurun_out/t_64x64_encoder_layers.txt) <(cut -c15-80 gpurun_out/t_def_encoder_layers.txt) | sed -n 1,13p
