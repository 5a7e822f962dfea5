/*
 * libplacebo-hip — transfer-function constants shared by the host colour code
 * (values of the reference's src/colorspace.h:22-46: SMPTE ST.2084, ARIB
 * STD-B67, Panasonic V-Log, Sony S-Log).
 */
#ifndef PLH_COLORSPACE_PRIV_H_
#define PLH_COLORSPACE_PRIV_H_

#include <libplacebo/colorspace.h>

static const float PQ_M1 = 2610./4096 * 1./4,
                   PQ_M2 = 2523./4096 * 128,
                   PQ_C1 = 3424./4096,
                   PQ_C2 = 2413./4096 * 32,
                   PQ_C3 = 2392./4096 * 32;

static const float HLG_A = 0.17883277,
                   HLG_B = 0.28466892,
                   HLG_C = 0.55991073,
                   HLG_REF = 1000.0 / PL_COLOR_SDR_WHITE;

static const float VLOG_B = 0.00873,
                   VLOG_C = 0.241514,
                   VLOG_D = 0.598206;

static const float SLOG_A = 0.432699,
                   SLOG_B = 0.037584,
                   SLOG_C = 0.616596 + 0.03,
                   SLOG_P = 3.538813,
                   SLOG_Q = 0.030001,
                   SLOG_K2 = 155.0 / 219.0;

// BT.1886 with black lift: L = a * (V + b)^2.4
void plh_bt1886_params(float csp_min, float csp_max, float *a, float *b);
// HLG system gamma `y` and black-lift `b` for a given display range
void plh_hlg_params(float csp_min, float csp_max, float *y, float *b);

#endif // PLH_COLORSPACE_PRIV_H_
