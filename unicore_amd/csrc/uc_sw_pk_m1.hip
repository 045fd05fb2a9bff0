// uc_sw_pk_m1.hip — instantiates the packed 16-bit gapped DP kernel classes for MODE 1 (uc_sw_pk_impl.hpp).
#include "uc_sw_pk_impl.hpp"
namespace uc {
void launch_sw_pk_class_m1(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
    launch_sw_pk_class_mode<1>(G, R, a, n_tasks, s);
}
void preload_sw_pk_m1() { preload_sw_pk_mode<1>(); }
}  // namespace uc
