// Tile edge lengths that get a register-resident codelet kernel (edges AFTER the cv2-style
// odd front pad). Sources: SURVEY.md Appendix C — VGG 32/16/8/4/2, ResNet-56/110 and
// DenseNet 32/16/8, GoogLeNet 32/16/8, ResNet-50 56/28/14/7, U2-Net-p 36/18/9 (+10 via
// the odd pad) and its 320-crop family 40/20/10; 64 and 48 are the natural power-of-two /
// 3*2^k fillers; round 3 adds the remaining even edges below 64 whose odd part is 3 or 15 (6, 12, 24, 30, 60 - and with
// them 5, 11, 23, 29, 59 through the odd front pad), so that an --input_size other than 224 / 288 / 320 does not
// drop small maps to the cosine-matrix kernel.
#pragma once
#ifdef DCTS_DEV_FAST  // development builds: a handful of instantiations, seconds instead of minutes
#define DCTS_CODELET_SIZES(X) X(7) X(8) X(9) X(14) X(28) X(56)
#ifndef DCTS_SPLIT_TABLE
#define DCTS_SPLIT_TABLE(X) X(128, 32, 2) X(224, 28, 3)
#endif
#ifndef DCTS_FUSED_TABLE
#ifndef DCTS_DEV_M224
#define DCTS_DEV_M224 14
#define DCTS_DEV_L224 4
#endif
#define DCTS_FUSED_TABLE(X) X(128, 16, 3) X(224, 14, 4)
#endif
#ifndef DCTS_PIPE_TABLE
#define DCTS_PIPE_TABLE(X) X(128, 16, 3) X(224, DCTS_DEV_M224, DCTS_DEV_L224)
#endif
#else
#define DCTS_CODELET_SIZES(X) \
  X(2) X(4) X(6) X(7) X(8) X(9) X(10) X(12) X(14) X(16) X(18) X(20) X(24) X(28) X(30) X(32) X(36) X(40) X(48) X(56) X(60) X(64)
#endif
