// nsf_units.h -- which (K bins, H hidden width) pairs each kernel unit instantiates (X(K, H) lists).
// Every H: the dim-major MFMA training kernel (one-layer flows) and the two-dims-per-wave kernel of small multi-layer / VJP
// launches (H = 16, round 5: one layer's panels resident; H = 8 also the two-lane kernel).  H = 8 and 16: the pipelined posterior
// walk.  Hidden widths in between run zero-padded in the next of these (nsf_kernels.hip: compiled_H).
// The reference accepts any K / hidden_dim (src/flows/flows.py:51-60); its examples use K in {5, 9, 12, 15}, H = 8.
#pragma once
#define NSF_UNITS(U) U(0) U(1) U(2) U(3) U(4) U(5) U(6) U(7) U(8) U(9) U(10) U(11)
#define NSF_KH_0(X) X(9, 8)
#define NSF_KH_1(X) X(5, 8) X(12, 8)
#define NSF_KH_2(X) X(6, 8) X(15, 8)
#define NSF_KH_3(X) X(2, 8) X(3, 8) X(4, 8) X(7, 8)
#define NSF_KH_4(X) X(8, 8) X(10, 8) X(11, 8)
#define NSF_KH_5(X) X(13, 8) X(14, 8)
#define NSF_KH_6(X) X(16, 8)
#define NSF_KH_7(X) X(2, 4) X(3, 4) X(4, 4) X(5, 4) X(6, 4) X(7, 4) X(8, 4) X(9, 4)
#define NSF_KH_8(X) X(10, 4) X(11, 4) X(12, 4) X(13, 4) X(14, 4) X(15, 4) X(16, 4)
#define NSF_KH_9(X) X(2, 16) X(3, 16) X(4, 16) X(5, 16) X(6, 16) X(7, 16)
#define NSF_KH_10(X) X(8, 16) X(9, 16) X(10, 16) X(11, 16) X(12, 16)
#define NSF_KH_11(X) X(13, 16) X(14, 16) X(15, 16) X(16, 16)
