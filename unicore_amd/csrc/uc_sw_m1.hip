// uc_sw_m1.hip — instantiates the gapped DP kernel classes for MODE 1 (see uc_sw_impl.hpp).
#include "uc_sw_impl.hpp"
namespace uc {
void launch_sw_class_m1(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
    launch_sw_class_mode<1>(G, R, a, n_tasks, s);
}
}  // namespace uc
