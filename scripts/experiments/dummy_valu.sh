# builds: hipcc <HIPCC_FLAGS of ihmr_amd/hip.py> [-DCONV_DUMMY_VALU=32|64] ihmr_amd/csrc/ihmr_hip.hip -o build/ab/dv{0,32,64}.so
# prints the 3 x 3 256 -> 256 and the 1 x 1 1024 -> 256 layer at 14 x 14 (a Stream-K launch + its fix-up) of each build
for v in dv0 dv32 dv64; do IHMR_HIP_LIBRARY=$PWD/build/ab/$v.so scripts/prof_encoder_layers.sh $v > /dev/null; echo $v; sed -n 31p gpurun_out/${v}_encoder_layers.txt; sed -n 30p gpurun_out/${v}_encoder_layers.txt; done
