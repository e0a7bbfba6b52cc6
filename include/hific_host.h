/* Host-side (CPU) C-ABI of hific_amd: entropy-table construction for the EVALUATION path.
 *
 * The reference builds its rANS tables with `maths.pmf_to_quantized_cdf` (src/helpers/maths.py:5-73), an O(n^2)
 * pure-Python loop it marks "TODO: port to C++", called once per scale / channel from `build_tables`
 * (src/compression/prior_model.py:77-120, src/compression/hyperprior_model.py:42-105).  This is that port: the same
 * ryg_rans "steal from the smallest frequency > 1" normalisation, bit-for-bit (the tables are part of the .hfc
 * bitstream contract).  Plain C, no torch types, no device code; built with g++ into libhific_host.so.
 */
#ifndef HIFIC_HOST_H
#define HIFIC_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HIFIC_HOST_OK 0
#define HIFIC_HOST_ERR_ARG (-1)      /* precision < 8 or > 32, n < 2, negative / NaN probabilities, zero total */
#define HIFIC_HOST_ERR_STEAL (-5)    /* no symbol with frequency > 1 left to steal from (reference: `assert best_steal != -1`) */

/* maths.py:5-73.  pmf: n float32 probabilities (un-normalised, as the reference passes them);
 * cdf: n+1 int64 entries, cdf[0] = 0, cdf[n] = 1 << precision, strictly increasing wherever the reference's is. */
int hific_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int64_t* cdf);

/* The per-row loop of `build_tables` (prior_model.py:105-113 / hyperprior_model.py:87-94): for each of `rows` pmfs
 * (row r: `lengths[r]` leading entries of pmf[r*stride ...]) append one extra mass - `extra[r]` (prior model: the tail
 * mass 2*lower[:,0]) or, when `extra` is NULL, max(0, 1 - sum(pmf_row)) (hyperprior model: the overflow mass) -
 * quantise, and write the row zero-padded to `cdf_stride` int32 entries (cdf_stride >= max(lengths) + 2). */
int hific_build_cdf_rows(const float* pmf, int rows, int stride, const int32_t* lengths, const float* extra,
                         int precision, int32_t* cdf, int cdf_stride);

const char* hific_host_version(void);

#ifdef __cplusplus
}
#endif
#endif
