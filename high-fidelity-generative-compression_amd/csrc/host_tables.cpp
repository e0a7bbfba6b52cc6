// CPU port of the reference's quantised-CDF construction (see include/hific_host.h).  Bit-exactness notes, each
// checked against the imported reference in tests/test_host_tables.py:
//  * torch.cumsum on a float32 CPU tensor accumulates in double and rounds every prefix to float32
//  * `cdf * target_total / empirical_total` is two float32 operations (the power-of-two multiply is exact)
//  * torch.round is round-half-to-even (nearbyintf under the default rounding mode)
//  * the hyperprior's overflow mass `1 - torch.sum(pmf_)` is a float32 sum: torch's CPU sum of a contiguous float32
//    vector is a vectorised pairwise reduction, not restated here - the Python wrapper computes that one scalar per
//    row with torch and passes it in as `extra` whenever bit-compatibility with reference-built tables is required
#include "../../include/hific_host.h"
#include <cmath>
#include <vector>

extern "C" const char* hific_host_version(void) { return "hific_host 0.1 (pmf_to_quantized_cdf port)"; }

extern "C" int hific_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int64_t* cdf) {
    if (!pmf || !cdf || n < 2 || precision < 8 || precision > 32) return HIFIC_HOST_ERR_ARG;
    const int64_t target_total = (int64_t)1 << precision;
    // prefix sums: double accumulator, float32 outputs (maths.py:31-33)
    std::vector<float> c((size_t)n + 1);
    c[0] = 0.f;
    double run = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!(pmf[i] >= 0.f)) return HIFIC_HOST_ERR_ARG;            // negative or NaN (maths.py:24-25)
        run += (double)pmf[i];
        c[(size_t)i + 1] = (float)run;
    }
    const float total = c[(size_t)n];
    if (!(total > 0.f)) return HIFIC_HOST_ERR_ARG;
    // normalise to the target precision (maths.py:36)
    const float tt = (float)target_total;
    for (int i = 0; i <= n; ++i) {
        const float scaled = (c[(size_t)i] * tt) / total;
        cdf[i] = (int64_t)nearbyintf(scaled);
    }
    // zero-frequency symbols steal one count from the smallest frequency > 1 (maths.py:41-64); first minimum wins
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        int64_t best_freq = target_total + 1;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const int64_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        if (best < 0) return HIFIC_HOST_ERR_STEAL;
        if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j] -= 1; }
        else          { for (int j = i + 1; j <= best; ++j) cdf[j] += 1; }
    }
    if (cdf[0] != 0 || cdf[n] != target_total) return HIFIC_HOST_ERR_ARG;
    return HIFIC_HOST_OK;
}

extern "C" int hific_build_cdf_rows(const float* pmf, int rows, int stride, const int32_t* lengths, const float* extra,
                                    int precision, int32_t* cdf, int cdf_stride) {
    if (!pmf || !lengths || !cdf || rows <= 0 || stride <= 0) return HIFIC_HOST_ERR_ARG;
    std::vector<float> row;
    std::vector<int64_t> q;
    for (int r = 0; r < rows; ++r) {
        const int len = lengths[r];
        if (len < 1 || len > stride || len + 2 > cdf_stride) return HIFIC_HOST_ERR_ARG;
        row.assign(pmf + (size_t)r * stride, pmf + (size_t)r * stride + len);
        float e;
        if (extra) e = extra[r];
        else {                                   // sequential float32 sum: see the note at the top of the file
            float s = 0.f;
            for (int i = 0; i < len; ++i) s += row[(size_t)i];
            e = 1.f - s; if (e < 0.f) e = 0.f;
        }
        row.push_back(e);
        q.resize((size_t)len + 2);
        const int rc = hific_pmf_to_quantized_cdf(row.data(), len + 1, precision, q.data());
        if (rc != HIFIC_HOST_OK) return rc;
        int32_t* out = cdf + (size_t)r * cdf_stride;
        for (int i = 0; i < len + 2; ++i) out[i] = (int32_t)q[(size_t)i];
        for (int i = len + 2; i < cdf_stride; ++i) out[i] = 0;
    }
    return HIFIC_HOST_OK;
}
