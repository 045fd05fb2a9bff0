// uc_sw_m3.hip — instantiates the gapped DP kernel classes for MODE 3 (traceback statistics, see uc_sw_impl.hpp).
#include "uc_sw_impl.hpp"
namespace uc {
void launch_sw_class_m3(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
    launch_sw_class_mode<3>(G, R, a, n_tasks, s);
}
}  // namespace uc
