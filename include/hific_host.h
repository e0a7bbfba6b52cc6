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

/* ---- rANS coder, scalar ("vectorize=False") path of the reference -----------------------
 * Bit-compatible port of `entropy_coding.ans_index_encoder` / `ans_index_decoder` (src/compression/entropy_coding.py:
 * 107-268, 479-559) over the 64-bit rANS of src/compression/ans.py:45-96: one coder state, 32-bit renormalisation
 * words, symbols outside [offset, offset + length - 2) sent as the overflow symbol followed by a 4-bit-nibble
 * variable-length code.  `symbols` / `indices` are the flattened int32 tensors the reference passes (the device
 * side produces them: hific_prior_symbols / hific_hyper_symbols); cdf is [rows][stride] uint32 with per-row
 * `cdf_length` / `cdf_offset` (the CDF / CDF_length / CDF_offset parameters of the entropy models).
 * The encoded message is the reference's `vrans.flatten` layout: [state >> 32, state & 0xffffffff, words...]. */
#define HIFIC_HOST_ERR_RANGE (-6)    /* index outside [0, rows), cdf_length outside [2, stride], precision outside [8,24] */
#define HIFIC_HOST_ERR_SPACE (-7)    /* output buffer too small: *out_len holds the required number of words */
#define HIFIC_HOST_ERR_DATA  (-8)    /* decoder ran out of words / corrupt message */
int hific_rans_encode(const int32_t* symbols, const int32_t* indices, long long n, const uint32_t* cdf, int rows,
                      int stride, const int32_t* cdf_length, const int32_t* cdf_offset, int precision,
                      uint32_t* out, long long out_cap, long long* out_len);
int hific_rans_decode(const uint32_t* encoded, long long enc_len, const int32_t* indices, long long n,
                      const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                      const int32_t* cdf_offset, int precision, int32_t* symbols);

/* ---- rANS coder, vectorised ("vectorize=True", the reference's default) path -------------------------------------
 * Bit-compatible port of `vec_ans_index_encoder` / `vec_ans_index_decoder` (entropy_coding.py:271-476, 561-673):
 * `lanes` interleaved coder states sharing one word stack, `steps` vector pushes.  The caller lays the int32
 * symbols / indices out as [steps][lanes] exactly as the reference does: batch 1 -> steps = H*W patches (PATCH_SIZE
 * (1,1), row-major), lanes = C channels (`compression_utils.decompose`); batch B > 1 -> steps = B, lanes = C*H*W.
 * Overflowing lanes are coded on a sub-stack of the masked lanes with 4-bit symbols.  The reference's quirks are
 * reproduced, because the bitstream depends on them: every nibble iteration re-pushes nibble 0 (`counter` is reset
 * inside the loop, :400/:642), so out-of-range symbols whose overflow code needs more than one nibble decode to the
 * value of their lowest nibble - with the reference's decoder and with this one alike - and lanes that have run out
 * of nibbles keep pushing their last value while any other lane still has some.  The branch for more than 14
 * nibbles (:382-391, "Undefined behaviour") is unreachable for int32 symbols and returns HIFIC_HOST_ERR_RANGE.
 * Message layout (`vrans.flatten`): [state >> 32 for every lane][state & 0xffffffff for every lane][words...]. */
int hific_rans_encode_vec(const int32_t* symbols, const int32_t* indices, long long steps, long long lanes,
                          const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                          const int32_t* cdf_offset, int precision, uint32_t* out, long long out_cap,
                          long long* out_len);
int hific_rans_decode_vec(const uint32_t* encoded, long long enc_len, const int32_t* indices, long long steps,
                          long long lanes, const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                          const int32_t* cdf_offset, int precision, int32_t* symbols);

const char* hific_host_version(void);

#ifdef __cplusplus
}
#endif
#endif
