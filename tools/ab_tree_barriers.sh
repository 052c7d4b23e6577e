B="--steps 20 --warmup 3 --no-cpu-baseline --concurrent 0 --no-slab-leg --no-bind-sweep"
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result"
bash tools/gpu.sh tests "msm or bullet or hyrax or golden or bit_exact_vs_oracle or slab_proof" r4j_tests
bash tools/gpu.sh bench r4j_new_a $B | grep -E "^r4j|msm opening"
$HC -DLASSO_TREE_BARRIERS -o lasso_amd/liblasso_hip.so lasso_amd/csrc/lasso_hip.hip 2>/dev/null
bash tools/gpu.sh bench r4j_old_a $B | grep -E "^r4j|msm opening"
bash tools/gpu.sh bench r4j_old_b $B | grep -E "^r4j|msm opening"
$HC -o lasso_amd/liblasso_hip.so lasso_amd/csrc/lasso_hip.hip 2>/dev/null
bash tools/gpu.sh bench r4j_new_b $B | grep -E "^r4j|msm opening"
bash tools/gpu.sh bench r4j_new_c $B | grep -E "^r4j|msm opening"
