// the fp64 oracle compiled as C++ with `real` = a scalar that rounds to fp32 inside selected stages (tests/tools/precision_study.py)
#include "round_real.hpp"
int g_round_on = 0;
unsigned g_round_mask = 0;
extern "C" {
#include "../../../oracle/mmo_engine.c"
void mmo_batch_rollout() {}   /* symbols the ctypes binding expects */
double mmo_test_seg_shape(int, const double*, const double*, const double*, double, double*, double*) { return 0; }
void mmo_round_mask(unsigned mask) { g_round_mask = mask; g_round_on = 0; }
}
