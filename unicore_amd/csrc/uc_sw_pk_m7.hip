// uc_sw_pk_m7.hip — instantiates the packed 16-bit gapped DP kernel classes for MODE 7 (traceback bytes of the box DP,
// uc_sw_pk_impl.hpp).
#include "uc_sw_pk_impl.hpp"
namespace uc {
void launch_sw_pk_class_m7(int G, int R, const SwArgs &a, uint32_t n_tasks, hipStream_t s) {
    launch_sw_pk_class_mode<7>(G, R, a, n_tasks, s);
}
}  // namespace uc
