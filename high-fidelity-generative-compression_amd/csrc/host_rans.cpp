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

inline bool check_tables(const int32_t* indices, long long n, int rows, int stride, const int32_t* cdf_length,
                         int precision) {
    if (precision < 8 || precision > 24 || rows <= 0 || stride < 2) return false;
    for (int r = 0; r < rows; ++r) if (cdf_length[r] < 2 || cdf_length[r] > stride) return false;
    for (long long i = 0; i < n; ++i) if (indices[i] < 0 || indices[i] >= rows) return false;
    return true;
}
// Exact h / f for the coder's operands (h < 2^63, 2 <= f <= 2^16) without a divide instruction: with m = floor(2^64 / f) + 1
// the estimate q' = floor(h * m / 2^64) satisfies q <= q' <= q + 1 (the error term h * (m - 2^64/f) / 2^64 is below 1/2), so
// one multiply-high, one multiply and a compare give the quotient and the remainder.  The table covers every frequency a
// 16-bit cdf (the reference's PRECISION_P) can hold and is built once per process; wider cdfs use the divide instruction.
struct Recip {
    std::vector<uint64_t> m;
    Recip() : m(((size_t)1 << 16) + 1, 0) {
        for (size_t f = 2; f < m.size(); ++f) m[f] = (uint64_t)((((unsigned __int128)1) << 64) / f) + 1;
    }
};
inline const uint64_t* recip_table(int precision) {
    if (precision > 16) return nullptr;
    static const Recip r;
    return r.m.data();
}
inline void divmod(uint64_t h, uint64_t f, const uint64_t* rcp, uint64_t& q, uint64_t& r) {
    if (f == 1) { q = h; r = 0; return; }
    if (!rcp) { q = h / f; r = h % f; return; }
    uint64_t qq = (uint64_t)(((unsigned __int128)h * rcp[f]) >> 64);
    uint64_t p = qq * f;
    if (p > h) { --qq; p -= f; }
    q = qq; r = h - p;
}
// Symbol search of the decoders: hint[row][cf >> (precision - 8)] = searchsorted(c[:len], bucket start, 'right') - 1,
// clamped to >= 0 - a position that is never past the answer, so a short walk from it ends at the same symbol as the
// reference's binary search (ans.py:80).
inline void build_hints(std::vector<uint16_t>& hint, const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                        int precision) {
    const int shift = precision - 8;
    hint.resize((size_t)rows * 256);
    for (int r = 0; r < rows; ++r) {
        const uint32_t* c = cdf + (size_t)r * stride;
        const int len = cdf_length[r];
        int s = 0;
        for (int b = 0; b < 256; ++b) {
            const uint64_t v = (uint64_t)b << shift;
            while (s + 1 < len && (uint64_t)c[s + 1] <= v) ++s;
            hint[(size_t)r * 256 + b] = (uint16_t)s;
        }
    }
}
// -> symbol s with c[s] <= cf < c[s + 1], or -1 when cf lies outside [c[0], c[len - 1]) (corrupt message)
inline int find_symbol(const uint32_t* c, int len, uint64_t cf, const uint16_t* hint_row, int shift) {
    if (hint_row) {
        if ((uint64_t)c[0] > cf) return -1;
        int s = hint_row[(size_t)(cf >> shift)];
        while (s + 1 < len && (uint64_t)c[s + 1] <= cf) ++s;
        return s + 1 >= len ? -1 : s;
    }
    int lo = 0, hi = len;                                       // searchsorted(c[:len], cf, 'right') - 1
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint64_t)c[mid] <= cf) lo = mid + 1; else hi = mid; }
    return (lo == 0 || lo >= len) ? -1 : lo - 1;
}
}  // namespace

extern "C" int hific_rans_encode(const int32_t* symbols, const int32_t* indices, long long n, const uint32_t* cdf,
                                 int rows, int stride, const int32_t* cdf_length, const int32_t* cdf_offset,
                                 int precision, uint32_t* out, long long out_cap, long long* out_len) {
    if (!symbols || !indices || !cdf || !cdf_length || !cdf_offset || !out_len || n < 0) return HIFIC_HOST_ERR_ARG;
    if (!check_tables(indices, n, rows, stride, cdf_length, precision)) return HIFIC_HOST_ERR_RANGE;
    // The reference records its pushes in symbol order and flushes them in reverse: walk the symbols backwards and flush
    // each symbol's own pushes (interval, then count nibbles, then value nibbles) last one first.
    uint64_t head = RANS_L;
    std::vector<uint32_t> words;                                    // oldest first
    words.reserve((size_t)n / 2 + 16);
    const uint64_t xm_sym = (RANS_L >> precision) << 32, xm_of = (RANS_L >> OVERFLOW_WIDTH) << 32;
    const uint64_t* rcp = recip_table(precision);
    uint32_t nib[64];                                               // 4-bit pushes of one symbol, in push order
    for (long long i = n; i-- > 0;) {
        const int row = indices[i];
        const uint32_t* c = cdf + (size_t)row * stride;
        const int64_t max_value = (int64_t)cdf_length[row] - 2;
        int64_t value = (int64_t)symbols[i] - (int64_t)cdf_offset[row];
        uint64_t overflow = 0;
        if (value < 0) { overflow = (uint64_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { overflow = (uint64_t)(2 * (value - max_value)); value = max_value; }
        const uint64_t start = c[value], freq = c[value + 1] - c[value];
        // zero-width interval, or a non-monotone / over-range row (freq wraps as unsigned): table not codable, and the
        // reciprocal table of divmod() only covers f <= 2^precision
        if (freq == 0 || freq > (1ull << precision)) return HIFIC_HOST_ERR_RANGE;
        if (value == max_value) {                                   // overflow symbol + nibble code (:213-238)
            int nn = 0;
            uint32_t widths = 0;
            while ((overflow >> (widths * OVERFLOW_WIDTH)) != 0) ++widths;
            uint32_t val = widths;
            while (val >= MAX_OVERFLOW) { nib[nn++] = MAX_OVERFLOW; val -= MAX_OVERFLOW; }
            nib[nn++] = val;
            for (uint32_t j = 0; j < widths; ++j) nib[nn++] = (uint32_t)((overflow >> (j * OVERFLOW_WIDTH)) & MAX_OVERFLOW);
            while (nn-- > 0) {                                      // freq 1: x_max = xm_of, h / 1 = h
                if (head >= xm_of) { words.push_back((uint32_t)head); head >>= 32; }
                head = (head << OVERFLOW_WIDTH) + nib[nn];
            }
        }
        if (head >= xm_sym * freq) { words.push_back((uint32_t)head); head >>= 32; }
        uint64_t q, r;
        divmod(head, freq, rcp, q, r);
        head = (q << precision) + r + start;
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
    std::vector<uint16_t> hint;
    const bool use_hint = stride <= 65535 && n >= 64ll * rows;        // building the table costs ~rows x (256 + len) steps
    if (use_hint) build_hints(hint, cdf, rows, stride, cdf_length, precision);
    const int shift = precision - 8;
    const uint16_t* hint_row = nullptr;
    // pop one symbol coded with `prec` bits against a sorted cdf of `len` entries (ans.py:74-96, :68-81)
    auto pop = [&](const uint32_t* c, int len, int prec, bool identity, uint32_t& sym) -> bool {
        const uint64_t cf = head & ((1ull << prec) - 1);
        uint32_t s, start, freq;
        if (identity) { s = (uint32_t)cf; start = s; freq = 1; }
        else {
            const int f = find_symbol(c, len, cf, hint_row, shift);
            if (f < 0) return false;
            s = (uint32_t)f; start = c[s]; freq = c[s + 1] - start;
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
        hint_row = use_hint ? hint.data() + (size_t)row * 256 : nullptr;
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
//
// Layout of the work (same bitstream as the reference, organised for a CPU instead of for numpy):
//   * the encoder visits the steps in reverse - the order in which the reference flushes its recorded pushes - so nothing
//     is recorded: a step's intervals are looked up, its (rare) overflow pushes built and flushed, then its vector push;
//   * the state update h -> (h / f << p) + h % f + start
//     uses a multiply-high by a tabulated reciprocal instead of a 64-bit divide (exact, see `divmod`): 2.6 ns per symbol.
//     (Lanes are independent coder states and could be flushed by several threads, stitching the word chunks afterwards;
//     measured on 8 cores it was 2.5x SLOWER than the single pass at 900k symbols - not kept.);
//   * the decoders get a cheaper symbol search: a per-row table of the cdf position at every 256th cumulative
//     frequency (precision >= 8), followed by a short linear walk (`find_symbol`).
// ---------------------------------------------------------------------------------------------------------------------

namespace {
struct OfPush {                       // one masked push of an overflow step: the participating lanes and their 4-bit symbols
    std::vector<uint32_t> lane, start;
};
}  // namespace

extern "C" int hific_rans_encode_vec(const int32_t* symbols, const int32_t* indices, long long steps, long long lanes,
                                     const uint32_t* cdf, int rows, int stride, const int32_t* cdf_length,
                                     const int32_t* cdf_offset, int precision, uint32_t* out, long long out_cap,
                                     long long* out_len) {
    if (!symbols || !indices || !cdf || !cdf_length || !cdf_offset || !out_len || steps < 0 || lanes <= 0)
        return HIFIC_HOST_ERR_ARG;
    if (!check_tables(indices, steps * lanes, rows, stride, cdf_length, precision)) return HIFIC_HOST_ERR_RANGE;
    const size_t L = (size_t)lanes, T = (size_t)steps;
    // The reference records its pushes in symbol order and flushes them in reverse (:448-466).  Here the steps are visited
    // in reverse directly; inside a step the flush order is: its overflow pushes, last one first, then its vector push.
    std::vector<uint64_t> head(L, RANS_L);
    std::vector<uint32_t> words;                                  // in emission order (oldest push first)
    std::vector<uint32_t> count;                                  // words emitted per push, in flush order
    words.reserve(T * L / 4 + 64);
    count.reserve(T + 64);
    std::vector<uint32_t> start(L), freq(L);
    std::vector<uint64_t> overflow(L), widths(L), val(L);
    std::vector<char> of_mask(L);
    std::vector<OfPush> pushes;
    const uint64_t xm_sym = (RANS_L >> precision) << 32, xm_of = (RANS_L >> OVERFLOW_WIDTH) << 32;
    const uint64_t* rcp = recip_table(precision);
    for (size_t t = T; t-- > 0;) {
        const int32_t* sym = symbols + t * L;
        const int32_t* idx = indices + t * L;
        bool any_of = false;
        for (size_t l = 0; l < L; ++l) {                          // intervals of the vector push (:300-345)
            const int row = idx[l];
            const uint32_t* c = cdf + (size_t)row * stride;
            const int64_t max_value = (int64_t)cdf_length[row] - 2;
            int64_t v = (int64_t)sym[l] - (int64_t)cdf_offset[row];
            uint64_t of = 0;
            if (v < 0) { of = (uint64_t)(-2 * v - 1); v = max_value; }
            else if (v >= max_value) { of = (uint64_t)(2 * (v - max_value)); v = max_value; }
            start[l] = c[v]; freq[l] = c[v + 1] - c[v];
            if (freq[l] == 0 || freq[l] > (1ull << precision)) return HIFIC_HOST_ERR_RANGE;   // not codable (see scalar coder)
            overflow[l] = of;
            of_mask[l] = (v == max_value);
            any_of |= of_mask[l] != 0;
        }
        if (any_of) {                                             // overflow pushes of this step, in push order
            uint64_t max_w = 0;
            for (size_t l = 0; l < L; ++l) {                      // widths of ALL lanes (overflow is 0 off the mask)
                uint64_t w = 0;
                while (w < 16 && (overflow[l] >> (w * OVERFLOW_WIDTH)) != 0) ++w;
                widths[l] = w; val[l] = w;
                if (w > max_w) max_w = w;
            }
            if (max_w >= MAX_OVERFLOW) return HIFIC_HOST_ERR_RANGE;   // the reference's "Undefined behaviour" branch
            pushes.clear();
            auto push_masked = [&]() {
                OfPush o;
                for (size_t l = 0; l < L; ++l) if (of_mask[l]) { o.lane.push_back((uint32_t)l); o.start.push_back((uint32_t)val[l]); }
                pushes.push_back(std::move(o));
            };
            push_masked();                                        // nibble counts (:393-398)
            for (uint64_t it = 0; it < max_w; ++it) {             // :400-413 - `counter` is reset every iteration
                for (size_t l = 0; l < L; ++l)
                    if (widths[l] != 0) { val[l] = overflow[l] & MAX_OVERFLOW; widths[l] -= 1; }
                push_masked();
            }
            for (size_t q = pushes.size(); q-- > 0;) {            // flushed last push first
                const OfPush& p = pushes[q];
                uint32_t n = 0;
                for (size_t j = 0; j < p.lane.size(); ++j) {
                    const size_t l = p.lane[j];
                    uint64_t h = head[l];
                    if (h >= xm_of) { words.push_back((uint32_t)h); h >>= 32; ++n; }           // freq 1: x_max = xm_of
                    head[l] = (h << OVERFLOW_WIDTH) + p.start[j];
                }
                count.push_back(n);
            }
        }
        uint32_t n = 0;
        for (size_t l = 0; l < L; ++l) {                          // the vector push itself (ans.py:45-72)
            uint64_t h = head[l];
            const uint64_t f = freq[l];
            if (h >= xm_sym * f) { words.push_back((uint32_t)h); h >>= 32; ++n; }
            uint64_t q, r;
            divmod(h, f, rcp, q, r);
            head[l] = (q << precision) + r + start[l];
        }
        count.push_back(n);
    }
    const long long need = 2 * lanes + (long long)words.size();
    *out_len = need;
    if (!out || out_cap < need) return HIFIC_HOST_ERR_SPACE;
    for (size_t l = 0; l < L; ++l) { out[l] = (uint32_t)(head[l] >> 32); out[L + l] = (uint32_t)head[l]; }
    // newest push first; inside a push the words keep their (lane) order
    size_t pos = 2 * L, end = words.size();
    for (size_t k = count.size(); k-- > 0;) {
        end -= count[k];
        for (uint32_t j = 0; j < count[k]; ++j) out[pos++] = words[end + j];
    }
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
    const int shift = precision - 8;
    std::vector<uint16_t> hint;
    const bool use_hint = stride <= 65535 && steps * lanes >= 64ll * rows;
    if (use_hint) build_hints(hint, cdf, rows, stride, cdf_length, precision);
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
    const uint64_t pmask = (1ull << precision) - 1;
    for (long long t = 0; t < steps; ++t) {
        const int32_t* idx = indices + (size_t)t * L;
        int32_t* out = symbols + (size_t)t * L;
        bool any_of = false;
        for (size_t l = 0; l < L; ++l) {                          // vector pop of the symbols (:606, ans.py:74-96)
            const int row = idx[l];
            const uint32_t* c = cdf + (size_t)row * stride;
            const int len = cdf_length[row];
            const uint64_t h = head[l];
            const uint64_t cf = h & pmask;
            const int s = find_symbol(c, len, cf, use_hint ? hint.data() + (size_t)row * 256 : nullptr, shift);
            if (s < 0) return HIFIC_HOST_ERR_DATA;
            const uint32_t st = c[s], fq = c[s + 1] - st;
            uint64_t nh = (uint64_t)fq * (h >> precision) + cf - st;
            // words go to the renormalising lanes in lane order: one pass is enough, a lane's word position depends only on
            // the lanes before it
            if (nh < RANS_L) { if (pos >= enc_len) return HIFIC_HOST_ERR_DATA; nh = (nh << 32) | enc[pos++]; }
            head[l] = nh;
            const int64_t mv = (int64_t)len - 2;
            max_value[l] = mv;
            value[l] = (int64_t)s;
            const bool of = (int64_t)s == mv;
            of_mask[l] = of;
            any_of |= of;
            out[l] = (int32_t)((int64_t)s + (int64_t)cdf_offset[row]);
        }
        if (!any_of) continue;
        if (!pop_masked()) return HIFIC_HOST_ERR_DATA;            // nibble counts
        for (size_t l = 0; l < L; ++l) if (of_mask[l]) {
            widths[l] = val[l]; overflow[l] = 0;
            // a count symbol of 15 would continue the count (:624-629); the encoder never emits it for int32 data
            if (val[l] == MAX_OVERFLOW) return HIFIC_HOST_ERR_DATA;
        }
        for (;;) {                                                // :634-645 - nibble 0 OR-ed once per iteration
            bool any_w = false;
            for (size_t l = 0; l < L; ++l) if (of_mask[l] && widths[l] != 0) any_w = true;
            if (!any_w) break;
            if (!pop_masked()) return HIFIC_HOST_ERR_DATA;
            for (size_t l = 0; l < L; ++l)
                if (of_mask[l] && widths[l] != 0) { overflow[l] |= val[l]; widths[l] -= 1; }
        }
        for (size_t l = 0; l < L; ++l) if (of_mask[l]) {          // :647-654
            int64_t v = (int64_t)(overflow[l] >> 1);
            if (overflow[l] & 1) v = -v - 1; else v += max_value[l];
            out[l] = (int32_t)(v + (int64_t)cdf_offset[idx[l]]);
        }
    }
    return HIFIC_HOST_OK;
}
