// lab_tables.h -- the two lookup tables behind the restated float32 RGB->Lab (devmath.h).
//
// f(x) is approximated on each segment [x_i, x_{i+1}) by the parabola through f(x_i), f(midpoint), f(x_{i+1}),
// stored as (a, b, c, x_i) with f ~ a + b*s + c*s^2, s = x - x_i.  Segment boundaries are the floats whose low
// 23-NB mantissa bits are zero, so the segment index and x_i come straight from the bit pattern of x.
// Coefficients are computed in float64 with libm and cast to float32; tests/test_abi_cpu.py checks that the
// tables equal the CPU oracle's (an independent build of the same definition) bit for bit.
#pragma once
#ifndef PYSP_LAB_DEC_NB
#define PYSP_LAB_DEC_NB 6                      // experiments only: the oracle's tables are built for 6
#endif
constexpr int LAB_DEC_NB = PYSP_LAB_DEC_NB, LAB_DEC_LOEXP = -5, LAB_DEC_N = 5 * (1 << LAB_DEC_NB) + 1;   // v in [2^-5, 1]
constexpr int LAB_CB_NB = 5, LAB_CB_LOEXP = -7, LAB_CB_N = 8 * (1 << LAB_CB_NB) + 1;       // t in [2^-7, 2)
// Host side: fills dec[LAB_DEC_N*4] and cb[LAB_CB_N*4] (segment order).
void host_lab_tables(float* dec, float* cb);
// Device layout: segment i of a table lives in slot (((127 + LOEXP) << NB) + i) & (SLOTS - 1), i.e. the slot is the
// bit field [23-NB, 23-NB+log2(SLOTS)) of the argument.  decode: 122<<6 = 7808 = 128 (mod 512), so segments 0..320
// occupy slots 128..448; cube root: 120<<5 = 3840 = 0 (mod 256), segments 0..255 occupy slots 0..255 (segment 256
// starts at t = 2 and is never addressed: X, Y, Z <= 1.0000001).  Unused slots are zero.
constexpr int LAB_DEC_SLOTS = 8 << LAB_DEC_NB, LAB_CB_SLOTS = 256, LAB_SLOTS = LAB_DEC_SLOTS + LAB_CB_SLOTS;   // 768 x 16 B = 12 KB
void host_lab_slots(float* slots /* LAB_SLOTS*4 */);
