// tools/emu: storage behind the kernels' dynamic LDS arrays (`extern __shared__ T name[]`) in the host functional model
#include <stdint.h>
namespace ksk {
thread_local unsigned long long s_test[64 * 1024 / 8];
thread_local unsigned long long s_bt[64 * 1024 / 8];
thread_local uint32_t s_tot[64 * 1024 / 4];
}  // namespace ksk
namespace ksrs {
thread_local uint32_t s_hist[64 * 1024 / 4];
}  // namespace ksrs
