// Scalar rANS coder of the reference, ported bit for bit (see include/hific_host.h).
//   state machine  : src/compression/ans.py:45-96   (RANS_L = 2^31, 64-bit state, 32-bit words)
//   symbol mapping : src/compression/entropy_coding.py:164-238 (encode), :506-557 (decode)
// The reference first records (start, freq, is_overflow) "instructions" in symbol order and then pushes them in
// reverse (LIFO) so that the decoder pops symbols in forward order; the message is head-first
// (`vrans.flatten`: state hi, state lo, then the renormalisation words newest first).
#include "../../include/hific_host.h"
#include <cstddef>
#include <cstdint>
#include <vector>

namespace {
constexpr uint64_t RANS_L = 1ull << 31;
constexpr int OVERFLOW_WIDTH = 4;                        // entropy_coding.py:8
constexpr uint32_t MAX_OVERFLOW = (1u << OVERFLOW_WIDTH) - 1;

struct Instr { uint32_t start, freq; bool overflow; };

inline bool check_tables(const int32_t* indices, long long n, int rows, int stride, const int32_t* cdf_length,
                         int precision) {
    if (precision < 8 || precision > 24 || rows <= 0 || stride < 2) return false;
    for (int r = 0; r < rows; ++r) if (cdf_length[r] < 2 || cdf_length[r] > stride) return false;
    for (long long i = 0; i < n; ++i) if (indices[i] < 0 || indices[i] >= rows) return false;
    return true;
}
}  // namespace

extern "C" int hific_rans_encode(const int32_t* symbols, const int32_t* indices, long long n, const uint32_t* cdf,
                                 int rows, int stride, const int32_t* cdf_length, const int32_t* cdf_offset,
                                 int precision, uint32_t* out, long long out_cap, long long* out_len) {
    if (!symbols || !indices || !cdf || !cdf_length || !cdf_offset || !out_len || n < 0) return HIFIC_HOST_ERR_ARG;
    if (!check_tables(indices, n, rows, stride, cdf_length, precision)) return HIFIC_HOST_ERR_RANGE;
    std::vector<Instr> ins;
    ins.reserve((size_t)n + 16);
    for (long long i = 0; i < n; ++i) {
        const int row = indices[i];
        const uint32_t* c = cdf + (size_t)row * stride;
        const int64_t max_value = (int64_t)cdf_length[row] - 2;
        int64_t value = (int64_t)symbols[i] - (int64_t)cdf_offset[row];
        uint64_t overflow = 0;
        if (value < 0) { overflow = (uint64_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { overflow = (uint64_t)(2 * (value - max_value)); value = max_value; }
        ins.push_back({c[value], c[value + 1] - c[value], false});
        if (value == max_value) {                                   // overflow symbol + nibble code (:213-238)
            uint32_t widths = 0;
            while ((overflow >> (widths * OVERFLOW_WIDTH)) != 0) ++widths;
            uint32_t val = widths;
            while (val >= MAX_OVERFLOW) { ins.push_back({MAX_OVERFLOW, 1u, true}); val -= MAX_OVERFLOW; }
            ins.push_back({val, 1u, true});
            for (uint32_t j = 0; j < widths; ++j)
                ins.push_back({(uint32_t)((overflow >> (j * OVERFLOW_WIDTH)) & MAX_OVERFLOW), 1u, true});
        }
    }
    // flush: push in reverse order (:241-258, ans.py:45-72)
    uint64_t head = RANS_L;
    std::vector<uint32_t> words;                                    // oldest first
    for (size_t k = ins.size(); k-- > 0;) {
        const Instr& in = ins[k];
        if (in.freq == 0) return HIFIC_HOST_ERR_RANGE;              // zero-width interval: table not codable
        const int prec = in.overflow ? OVERFLOW_WIDTH : precision;
        const uint64_t x_max = ((RANS_L >> prec) << 32) * (uint64_t)in.freq;
        if (head >= x_max) { words.push_back((uint32_t)head); head >>= 32; }
        head = ((head / in.freq) << prec) + (head % in.freq) + in.start;
    }
    const long long need = 2 + (long long)words.size();
    *out_len = need;
    if (!out || out_cap < need) return HIFIC_HOST_ERR_SPACE;
    out[0] = (uint32_t)(head >> 32); out[1] = (uint32_t)head;
    for (size_t k = 0; k < words.size(); ++k) out[2 + k] = words[words.size() - 1 - k];   // newest first
    return HIFIC_HOST_OK;
}

extern "C" int hific_rans_decode(const uint32_t* enc, long long enc_len, const int32_t* indices, long long n,
                                 const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                                 const int32_t* cdf_offset, int precision, int32_t* symbols) {
    if (!enc || !indices || !cdf || !cdf_length || !cdf_offset || !symbols || n < 0 || enc_len < 2)
        return HIFIC_HOST_ERR_ARG;
    if (!check_tables(indices, n, rows, stride, cdf_length, precision)) return HIFIC_HOST_ERR_RANGE;
    uint64_t head = ((uint64_t)enc[0] << 32) | enc[1];
    long long pos = 2;
    // pop one symbol coded with `prec` bits against a sorted cdf of `len` entries (ans.py:74-96, :68-81)
    auto pop = [&](const uint32_t* c, int len, int prec, bool identity, uint32_t& sym) -> bool {
        const uint64_t cf = head & ((1ull << prec) - 1);
        uint32_t s, start, freq;
        if (identity) { s = (uint32_t)cf; start = s; freq = 1; }
        else {
            int lo = 0, hi = len;                                   // searchsorted(c[:len], cf, 'right') - 1
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint64_t)c[mid] <= cf) lo = mid + 1; else hi = mid; }
            if (lo == 0 || lo >= len) return false;
            s = (uint32_t)(lo - 1); start = c[s]; freq = c[s + 1] - start;
        }
        head = (uint64_t)freq * (head >> prec) + cf - start;
        if (head < RANS_L) {
            if (pos >= enc_len) return false;
            head = (head << 32) | enc[pos++];
        }
        sym = s;
        return true;
    };
    for (long long i = 0; i < n; ++i) {
        const int row = indices[i];
        const uint32_t* c = cdf + (size_t)row * stride;
        const int len = cdf_length[row];
        const int64_t max_value = (int64_t)len - 2;
        uint32_t s;
        if (!pop(c, len, precision, false, s)) return HIFIC_HOST_ERR_DATA;
        int64_t value = (int64_t)s;
        if (value == max_value) {                                   // :527-551
            uint32_t val;
            if (!pop(nullptr, 0, OVERFLOW_WIDTH, true, val)) return HIFIC_HOST_ERR_DATA;
            uint64_t widths = val;
            while (val == MAX_OVERFLOW) {
                if (!pop(nullptr, 0, OVERFLOW_WIDTH, true, val)) return HIFIC_HOST_ERR_DATA;
                widths += val;
            }
            if (widths > 16) return HIFIC_HOST_ERR_DATA;
            uint64_t overflow = 0;
            for (uint64_t j = 0; j < widths; ++j) {
                if (!pop(nullptr, 0, OVERFLOW_WIDTH, true, val)) return HIFIC_HOST_ERR_DATA;
                overflow |= (uint64_t)val << (j * OVERFLOW_WIDTH);
            }
            value = (int64_t)(overflow >> 1);
            if (overflow & 1) value = -value - 1; else value += max_value;
        }
        symbols[i] = (int32_t)(value + (int64_t)cdf_offset[row]);
    }
    return HIFIC_HOST_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Vectorised path: entropy_coding.py:271-476 (encode), :561-673 (decode), ans.py:45-96 on arrays of lanes.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct VInstr {                       // one vector push: all lanes (mask empty) or the masked lanes of an overflow step
    bool overflow;
    std::vector<uint32_t> start, freq;      // per participating lane, lane order
    std::vector<uint32_t> lane;             // participating lanes (overflow pushes only)
};
}  // namespace

extern "C" int hific_rans_encode_vec(const int32_t* symbols, const int32_t* indices, long long steps, long long lanes,
                                     const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                                     const int32_t* cdf_offset, int precision, uint32_t* out, long long out_cap,
                                     long long* out_len) {
    if (!symbols || !indices || !cdf || !cdf_length || !cdf_offset || !out_len || steps < 0 || lanes <= 0)
        return HIFIC_HOST_ERR_ARG;
    if (!check_tables(indices, steps * lanes, rows, stride, cdf_length, precision)) return HIFIC_HOST_ERR_RANGE;
    const size_t L = (size_t)lanes;
    std::vector<VInstr> ins;
    ins.reserve((size_t)steps * 2);
    std::vector<int64_t> value(L), max_value(L);
    std::vector<uint64_t> overflow(L), widths(L), val(L);
    std::vector<char> of_mask(L);
    for (long long t = 0; t < steps; ++t) {
        const int32_t* sym = symbols + (size_t)t * L;
        const int32_t* idx = indices + (size_t)t * L;
        VInstr vi; vi.overflow = false; vi.start.resize(L); vi.freq.resize(L);
        bool any_of = false;
        for (size_t l = 0; l < L; ++l) {
            const int row = idx[l];
            const uint32_t* c = cdf + (size_t)row * stride;
            max_value[l] = (int64_t)cdf_length[row] - 2;
            int64_t v = (int64_t)sym[l] - (int64_t)cdf_offset[row];
            overflow[l] = 0;
            if (v < 0) { overflow[l] = (uint64_t)(-2 * v - 1); v = max_value[l]; }
            else if (v >= max_value[l]) { overflow[l] = (uint64_t)(2 * (v - max_value[l])); v = max_value[l]; }
            value[l] = v;
            vi.start[l] = c[v]; vi.freq[l] = c[v + 1] - c[v];
            if (vi.freq[l] == 0) return HIFIC_HOST_ERR_RANGE;
            of_mask[l] = (v == max_value[l]);
            any_of |= of_mask[l] != 0;
        }
        ins.push_back(std::move(vi));
        if (!any_of) continue;
        uint64_t max_w = 0;
        for (size_t l = 0; l < L; ++l) {                          // widths of ALL lanes (overflow is 0 off the mask)
            uint64_t w = 0;
            while (w < 16 && (overflow[l] >> (w * OVERFLOW_WIDTH)) != 0) ++w;
            widths[l] = w; val[l] = w;
            if (w > max_w) max_w = w;
        }
        if (max_w >= MAX_OVERFLOW) return HIFIC_HOST_ERR_RANGE;   // the reference's "Undefined behaviour" branch
        auto push_masked = [&]() {
            VInstr o; o.overflow = true;
            for (size_t l = 0; l < L; ++l) if (of_mask[l]) { o.lane.push_back((uint32_t)l); o.start.push_back((uint32_t)val[l]); o.freq.push_back(1u); }
            ins.push_back(std::move(o));
        };
        push_masked();                                            // nibble counts (:393-398)
        for (uint64_t it = 0; it < max_w; ++it) {                 // :400-413 - `counter` is reset every iteration
            for (size_t l = 0; l < L; ++l)
                if (widths[l] != 0) { val[l] = overflow[l] & MAX_OVERFLOW; widths[l] -= 1; }
            push_masked();
        }
    }
    // flush in reverse (:448-466)
    std::vector<uint64_t> head(L, RANS_L);
    std::vector<std::vector<uint32_t>> chunks;                   // oldest first
    for (size_t k = ins.size(); k-- > 0;) {
        const VInstr& in = ins[k];
        const int prec = in.overflow ? OVERFLOW_WIDTH : precision;
        const size_t n = in.start.size();
        std::vector<uint32_t> chunk;
        for (size_t j = 0; j < n; ++j) {
            const size_t l = in.overflow ? in.lane[j] : j;
            const uint64_t x_max = ((RANS_L >> prec) << 32) * (uint64_t)in.freq[j];
            if (head[l] >= x_max) { chunk.push_back((uint32_t)head[l]); head[l] >>= 32; }
        }
        if (!chunk.empty()) chunks.push_back(std::move(chunk));
        for (size_t j = 0; j < n; ++j) {
            const size_t l = in.overflow ? in.lane[j] : j;
            head[l] = ((head[l] / in.freq[j]) << prec) + (head[l] % in.freq[j]) + in.start[j];
        }
    }
    long long need = 2 * lanes;
    for (const auto& c : chunks) need += (long long)c.size();
    *out_len = need;
    if (!out || out_cap < need) return HIFIC_HOST_ERR_SPACE;
    for (size_t l = 0; l < L; ++l) { out[l] = (uint32_t)(head[l] >> 32); out[L + l] = (uint32_t)head[l]; }
    size_t pos = 2 * L;
    for (size_t k = chunks.size(); k-- > 0;) for (uint32_t w : chunks[k]) out[pos++] = w;     // newest chunk first
    return HIFIC_HOST_OK;
}

extern "C" int hific_rans_decode_vec(const uint32_t* enc, long long enc_len, const int32_t* indices, long long steps,
                                     long long lanes, const uint32_t* cdf, int rows, int stride,
                                     const int32_t* cdf_length, const int32_t* cdf_offset, int precision,
                                     int32_t* symbols) {
    if (!enc || !indices || !cdf || !cdf_length || !cdf_offset || !symbols || steps < 0 || lanes <= 0 ||
        enc_len < 2 * lanes)
        return HIFIC_HOST_ERR_ARG;
    if (!check_tables(indices, steps * lanes, rows, stride, cdf_length, precision)) return HIFIC_HOST_ERR_RANGE;
    const size_t L = (size_t)lanes;
    std::vector<uint64_t> head(L);
    for (size_t l = 0; l < L; ++l) head[l] = ((uint64_t)enc[l] << 32) | enc[L + l];
    long long pos = 2 * lanes;
    std::vector<int64_t> value(L), max_value(L);
    std::vector<uint64_t> widths(L), overflow(L), val(L);
    std::vector<char> of_mask(L), renorm(L);
    // pop 4-bit symbols from the masked lanes (substack, :432-444): symbol = cf, start = cf, freq = 1
    auto pop_masked = [&]() -> bool {
        for (size_t l = 0; l < L; ++l) {
            renorm[l] = 0;
            if (!of_mask[l]) continue;
            const uint64_t cf = head[l] & MAX_OVERFLOW;
            val[l] = cf;
            head[l] = head[l] >> OVERFLOW_WIDTH;
            renorm[l] = head[l] < RANS_L;
        }
        for (size_t l = 0; l < L; ++l)
            if (renorm[l]) { if (pos >= enc_len) return false; head[l] = (head[l] << 32) | enc[pos++]; }
        return true;
    };
    for (long long t = 0; t < steps; ++t) {
        const int32_t* idx = indices + (size_t)t * L;
        bool any_of = false;
        for (size_t l = 0; l < L; ++l) {                          // vector pop of the symbols (:606, ans.py:74-96)
            const int row = idx[l];
            const uint32_t* c = cdf + (size_t)row * stride;
            const int len = cdf_length[row];
            max_value[l] = (int64_t)len - 2;
            const uint64_t cf = head[l] & ((1ull << precision) - 1);
            int lo = 0, hi = len;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint64_t)c[mid] <= cf) lo = mid + 1; else hi = mid; }
            if (lo == 0 || lo >= len) return HIFIC_HOST_ERR_DATA;
            const uint32_t s = (uint32_t)(lo - 1), start = c[s], freq = c[s + 1] - start;
            head[l] = (uint64_t)freq * (head[l] >> precision) + cf - start;
            renorm[l] = head[l] < RANS_L;
            value[l] = (int64_t)s;
            of_mask[l] = value[l] == max_value[l];
            any_of |= of_mask[l] != 0;
        }
        for (size_t l = 0; l < L; ++l)                            // words go to the renormalising lanes in lane order
            if (renorm[l]) { if (pos >= enc_len) return HIFIC_HOST_ERR_DATA; head[l] = (head[l] << 32) | enc[pos++]; }
        if (any_of) {
            if (!pop_masked()) return HIFIC_HOST_ERR_DATA;        // nibble counts
            for (size_t l = 0; l < L; ++l) if (of_mask[l]) {
                widths[l] = val[l]; overflow[l] = 0;
                // a count symbol of 15 would continue the count (:624-629); the encoder never emits it for int32 data
                if (val[l] == MAX_OVERFLOW) return HIFIC_HOST_ERR_DATA;
            }
            for (;;) {                                            // :634-645 - nibble 0 OR-ed once per iteration
                bool any_w = false;
                for (size_t l = 0; l < L; ++l) if (of_mask[l] && widths[l] != 0) any_w = true;
                if (!any_w) break;
                if (!pop_masked()) return HIFIC_HOST_ERR_DATA;
                for (size_t l = 0; l < L; ++l)
                    if (of_mask[l] && widths[l] != 0) { overflow[l] |= val[l]; widths[l] -= 1; }
            }
            for (size_t l = 0; l < L; ++l) if (of_mask[l]) {      // :647-654
                int64_t v = (int64_t)(overflow[l] >> 1);
                if (overflow[l] & 1) v = -v - 1; else v += max_value[l];
                value[l] = v;
            }
        }
        int32_t* out = symbols + (size_t)t * L;
        for (size_t l = 0; l < L; ++l) out[l] = (int32_t)(value[l] + (int64_t)cdf_offset[idx[l]]);
    }
    return HIFIC_HOST_OK;
}
