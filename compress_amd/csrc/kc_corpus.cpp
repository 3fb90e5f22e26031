// kc_corpus.cpp — deterministic synthetic corpora for benchmarks and parity tests
// (SURVEY.md §8d): every unit is generated independently from splitmix64(seed ^ unit_index),
// so generation is embarrassingly parallel over host threads and any sub-range of units can be
// regenerated bit-identically.  Integer arithmetic only (no libm) so the bytes do not depend on
// the platform's math library.
//   'T' enwik-style text : Zipf(s=1) words from a 50 000-word synthetic vocabulary, sentence and
//                          paragraph punctuation, [[wiki links]], XML-ish page headers, numbers
//   'H' high entropy     : raw PRNG bytes
//   'J' JSON records     : newline-delimited records with Zipf-distributed strings
//   'M' mixed            : 32 KiB segments cycling T / H / small-delta u32 arrays / zero pages
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>
#include <algorithm>
#include "../../include/kcgpu.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    inline uint64_t next() {  // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    inline uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

struct Vocab {
    std::vector<std::string> words;
    std::vector<uint64_t> cdf;  // cumulative Zipf(s=1) weights
    uint64_t total = 0;
};

const Vocab& vocab() {
    static Vocab v;
    static std::once_flag once;
    std::call_once(once, [] {
        // letter frequencies (per 1000) roughly English
        static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
        static const int freq[] = {127, 91, 82, 75, 70, 67, 63, 61, 60, 43, 40, 28, 28, 24, 24, 22, 20, 20, 19, 15, 10, 8, 2, 2, 1, 1};
        std::vector<int> cum;
        int tot = 0;
        for (int f : freq) { tot += f; cum.push_back(tot); }
        Rng r(0x5EEDC0DEULL);
        const int N = 50000;
        v.words.reserve(N);
        for (int i = 0; i < N; i++) {
            // frequent words are short: length grows with rank
            int minLen = i < 64 ? 1 : (i < 1024 ? 2 : 3);
            int span = i < 64 ? 3 : (i < 1024 ? 5 : 9);
            int len = minLen + (int)r.below((uint32_t)span + 1);
            std::string w;
            for (int k = 0; k < len; k++) {
                int x = (int)r.below((uint32_t)tot);
                int li = (int)(std::upper_bound(cum.begin(), cum.end(), x) - cum.begin());
                w.push_back(letters[li]);
            }
            v.words.push_back(w);
        }
        v.cdf.resize(N);
        uint64_t acc = 0;
        for (int i = 0; i < N; i++) {
            acc += (uint64_t)0x100000000ULL / (uint64_t)(i + 1) + (i < 2048 ? (uint64_t)0x400000000ULL / (uint64_t)(i + 6) : 0);
            v.cdf[i] = acc;
        }
        v.total = acc;
    });
    return v;
}

inline const std::string& zipf_word(Rng& r) {
    const Vocab& v = vocab();
    // 64-bit uniform in [0,total): multiply-high
    unsigned __int128 m = (unsigned __int128)r.next() * (unsigned __int128)v.total;
    uint64_t x = (uint64_t)(m >> 64);
    size_t i = (size_t)(std::upper_bound(v.cdf.begin(), v.cdf.end(), x) - v.cdf.begin());
    if (i >= v.words.size()) i = v.words.size() - 1;
    return v.words[i];
}

struct Out {
    uint8_t* p;
    size_t cap, n;
    inline bool full() const { return n >= cap; }
    inline void put(char c) { if (n < cap) p[n] = (uint8_t)c; n++; }
    inline void puts(const char* s) { while (*s) put(*s++); }
    inline void puts(const std::string& s) { for (char c : s) put(c); }
    inline void putnum(uint64_t v) {
        char b[24];
        int k = 0;
        do { b[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) put(b[--k]);
    }
};

void gen_text(Rng& r, Out& o) {
    const std::string* ring[512];  // recent words: topical phrase repetition, as in encyclopedic text
    uint32_t ringN = 0, phraseLeft = 0, phrasePos = 0;
    uint32_t sinceHeader = 4096;  // force a header at the start
    uint32_t wordsInSentence = 0, sentLen = 8 + r.below(18);
    bool cap = true;
    while (!o.full()) {
        if (sinceHeader >= 4096) {
            size_t s0 = o.n;
            o.puts("<page>\n  <title>");
            int tw = 1 + (int)r.below(4);
            for (int k = 0; k < tw; k++) { if (k) o.put(' '); std::string w = zipf_word(r); w[0] = (char)(w[0] - 32); o.puts(w); }
            o.puts("</title>\n  <id>");
            o.putnum(r.below(9000000) + 1000);
            o.puts("</id>\n  <revision>\n    <timestamp>20");
            o.putnum(10 + r.below(15));
            o.put('-'); o.putnum(10 + r.below(3)); o.put('-'); o.putnum(10 + r.below(19));
            o.puts("T"); o.putnum(10 + r.below(14)); o.put(':'); o.putnum(10 + r.below(50)); o.put(':'); o.putnum(10 + r.below(50));
            o.puts("Z</timestamp>\n    <text xml:space=\"preserve\">");
            sinceHeader = 0;
            cap = true;
            (void)s0;
        }
        size_t before = o.n;
        const uint32_t roll = r.below(1000);
        if (roll < 25) {  // [[wiki link]]
            o.puts("[[");
            int tw = 1 + (int)r.below(3);
            for (int k = 0; k < tw; k++) { if (k) o.put(' '); o.puts(zipf_word(r)); }
            if (r.below(3) == 0) { o.put('|'); o.puts(zipf_word(r)); }
            o.puts("]]");
        } else if (roll < 40) {  // number
            o.putnum(r.below(3000));
        } else if (roll < 46) {
            o.puts("''"); o.puts(zipf_word(r)); o.puts("''");
        } else {
            if (phraseLeft == 0 && ringN >= 64 && r.below(100) < 14) {
                phraseLeft = 2 + r.below(5);
                phrasePos = ringN - 8 - r.below(ringN < 512 ? ringN - 8 : 504);
            }
            const std::string* wp;
            if (phraseLeft > 0) { wp = ring[phrasePos & 511]; phrasePos++; phraseLeft--; }
            else wp = &zipf_word(r);
            ring[ringN & 511] = wp;
            ringN++;
            const std::string& w = *wp;
            if (cap) { o.put((char)(w[0] - 32)); for (size_t k = 1; k < w.size(); k++) o.put(w[k]); cap = false; }
            else o.puts(w);
        }
        wordsInSentence++;
        if (wordsInSentence >= sentLen) {
            o.put(r.below(10) == 0 ? '?' : '.');
            wordsInSentence = 0;
            sentLen = 6 + r.below(22);
            cap = true;
            if (r.below(6) == 0) { o.puts("\n\n"); if (r.below(5) == 0) { o.puts("== "); o.puts(zipf_word(r)); o.puts(" ==\n"); } }
            else o.put(' ');
        } else {
            if (r.below(12) == 0) o.put(',');
            o.put(' ');
        }
        sinceHeader += (uint32_t)(o.n - before);
        if (sinceHeader >= 4096) o.puts("</text>\n    </revision>\n</page>\n");
    }
}

void gen_entropy(Rng& r, Out& o) {
    while (o.n + 8 <= o.cap) { uint64_t v = r.next(); memcpy(o.p + o.n, &v, 8); o.n += 8; }
    while (o.n < o.cap) o.put((char)r.next());
}

void gen_json(Rng& r, Out& o) {
    static const char* events[] = {"PushEvent", "PullRequestEvent", "IssuesEvent", "WatchEvent", "ForkEvent", "CreateEvent", "IssueCommentEvent", "DeleteEvent"};
    uint64_t id = 2489651045ULL + r.below(1000000);
    uint64_t ts = 1420070400ULL + r.below(100000);
    while (!o.full()) {
        o.puts("{\"id\":\""); o.putnum(id); id += 1 + r.below(40);
        o.puts("\",\"type\":\""); o.puts(events[r.below(8) < 4 ? 0 : r.below(8)]);
        o.puts("\",\"actor\":{\"id\":"); o.putnum(r.below(9000000));
        o.puts(",\"login\":\""); o.puts(zipf_word(r)); o.putnum(r.below(100));
        o.puts("\",\"gravatar_id\":\"\",\"url\":\"https://api.github.com/users/"); o.puts(zipf_word(r));
        o.puts("\"},\"repo\":{\"id\":"); o.putnum(r.below(30000000));
        o.puts(",\"name\":\""); o.puts(zipf_word(r)); o.put('/'); o.puts(zipf_word(r));
        o.puts("\"},\"payload\":{\"push_id\":"); o.putnum(536740000ULL + r.below(1000000));
        o.puts(",\"size\":"); o.putnum(1 + r.below(3));
        o.puts(",\"ref\":\"refs/heads/"); o.puts(r.below(3) ? "master" : zipf_word(r).c_str());
        o.puts("\",\"tags\":[");
        int nt = (int)r.below(4);
        for (int k = 0; k < nt; k++) { if (k) o.put(','); o.put('"'); o.puts(zipf_word(r)); o.put('"'); }
        o.puts("],\"score\":"); o.putnum(r.below(1000)); o.put('.'); o.putnum(r.below(100));
        o.puts(",\"msg\":\"");
        int nw = 3 + (int)r.below(14);
        for (int k = 0; k < nw; k++) { if (k) o.put(' '); o.puts(zipf_word(r)); }
        o.puts("\"},\"public\":true,\"created_at\":\"2015-01-01T"); ts += r.below(3);
        o.putnum(10 + (ts / 3600) % 14); o.put(':'); o.putnum(10 + (ts / 60) % 50); o.put(':'); o.putnum(10 + ts % 50);
        o.puts("Z\"}\n");
    }
}

void gen_deltas(Rng& r, Out& o) {
    uint32_t v = (uint32_t)r.next();
    while (o.n + 4 <= o.cap) { v += r.below(16); memcpy(o.p + o.n, &v, 4); o.n += 4; }
    while (o.n < o.cap) o.put(0);
}

void gen_unit(int kind, uint64_t seed, uint64_t unit_index, uint8_t* dst, uint32_t size) {
    Rng r(seed ^ (unit_index * 0xD1B54A32D192ED03ULL + 0x9E3779B97F4A7C15ULL));
    r.next();
    Out o{dst, size, 0};
    switch (kind) {
    case 'T': gen_text(r, o); break;
    case 'H': gen_entropy(r, o); break;
    case 'J': gen_json(r, o); break;
    case 'M': {
        const uint32_t seg = 32 << 10;
        for (uint32_t off = 0, k = 0; off < size; off += seg, k++) {
            Out so{dst + off, std::min(seg, size - off), 0};
            switch ((unit_index + k) & 3) {
            case 0: gen_text(r, so); break;
            case 1: gen_entropy(r, so); break;
            case 2: gen_deltas(r, so); break;
            default: memset(so.p, 0, so.cap); break;
            }
        }
        break;
    }
    default: memset(dst, 0, size); break;
    }
}

}  // namespace

extern "C" kc_status kc_corpus_fill(int kind, uint64_t seed, uint64_t first_unit, uint32_t n_units, uint32_t unit_size, uint8_t* dst, int threads) {
    if (!dst || unit_size == 0) return KC_ERR_BAD_ARG;
    if (kind != 'T' && kind != 'H' && kind != 'J' && kind != 'M') return KC_ERR_BAD_ARG;
    if (threads < 1) threads = 1;
    vocab();
    std::atomic<uint32_t> next(0);
    auto worker = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= n_units) break;
            gen_unit(kind, seed, first_unit + i, dst + (size_t)i * unit_size, unit_size);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return KC_OK;
}
