// Experiment build of the conv-encoder kernels (tools/exp/roles_ab.py): the product source plus the round-2 backward kernel
// and the exp_* entry points, which live here and not under unreal_amd/csrc/.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/exp/encoder_ablate.hip -o tools/exp/build/libenc_ablate.so
#define UNREAL_ABLATE 1
#define UNREAL_EXP_KERNELS "../../tools/exp/encoder_bwd_r2.inc"
#define UNREAL_EXP_ENTRIES "../../tools/exp/encoder_exp_entries.inc"
#include "../../unreal_amd/csrc/encoder.hip"
