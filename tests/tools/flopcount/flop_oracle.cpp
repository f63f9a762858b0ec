// the fp64 oracle compiled as C++ with `real` = an operation-counting scalar (tests/tools/count_flops.py)
#include "count_real.hpp"
FlopCounters g_fc = {0, 0, 0, 0, 0, 0};
extern "C" {
#include "../../../oracle/mmo_engine.c"
void mmo_batch_rollout() {}   /* symbols the ctypes binding expects; not used by the counter */
double mmo_test_seg_shape(int, const double*, const double*, const double*, double, double*, double*) { return 0; }   /* symbol the binding expects */
void mmo_flops_reset() { g_fc = FlopCounters{0, 0, 0, 0, 0, 0}; }
void mmo_flops_get(uint64_t* out) { out[0] = g_fc.add; out[1] = g_fc.mul; out[2] = g_fc.div; out[3] = g_fc.sqrt_; out[4] = g_fc.trans; out[5] = g_fc.cmp; }
}
