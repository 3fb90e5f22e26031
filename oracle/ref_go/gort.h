// gort.h — Go semantics for the C++ that go2cpp.py writes from the reference's Go sources.  TEST INFRASTRUCTURE (oracle/).
//
// The translation is syntax-directed; this header is what makes the C++ compiler check and run Go's rules:
//   * I<T>: fixed-width integers with Go's arithmetic (wrap-around, no implicit widening, operands of ONE type, shifts by any
//     unsigned count with counts >= the width giving 0 / the sign), K: untyped integer constants (128-bit, adapts to the other
//     operand, becomes `int` when a variable is defined from it);
//   * Slice<T> / Array<T,N> / String: Go's slices (shared backing store, len / cap, append growth irrelevant to the bytes), value
//     arrays, bounds checks that panic;
//   * error, panic, defer, sync.Once, interfaces by type erasure (the translator writes one adapter class per interface).
// Memory: there is no collector; what a call of the driver allocates (make, new, append growth, escaping locals) is logged by
// go::rt and released when the call ends (rt::Scope), large blocks going to a size-keyed cache so that the next call's tables are
// cleared in place instead of faulted in page by page; package initialisation runs with the log off (rt::Permanent).
// Standard-library pieces the sources call (math/bits,
// encoding/binary, bytes.Equal, math.Log2 ...) are restated at the end: they are the Go standard library's, not the reference's.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace go {
namespace rt {
struct Blk { void* p; size_t n; };
struct Log { std::vector<Blk> blocks; };
inline Log*& cur() { static thread_local Log* c = nullptr; return c; }
struct Cache {
    std::mutex mu;
    std::multimap<size_t, void*> free_;  // blocks of >= BIG bytes released by finished calls, by exact size
    size_t held = 0;
    static Cache& get() { static Cache* c = new Cache(); return *c; }
};
constexpr size_t BIG = (size_t)256 << 10, CACHE_MAX = (size_t)1 << 30;
// zeroed memory that lives until the running call ends (or for ever: package initialisation, calls outside a Scope)
inline void* alloc(size_t n) {
    if (n == 0) n = 1;
    Log* l = cur();
    void* p = nullptr;
    if (l != nullptr && n >= BIG) {
        Cache& c = Cache::get();
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.free_.find(n);
        if (it != c.free_.end()) { p = it->second; c.free_.erase(it); c.held -= n; }
    }
    if (p != nullptr) memset(p, 0, n);
    else p = calloc(1, n);
    if (p == nullptr) throw std::bad_alloc();
    if (l != nullptr) l->blocks.push_back(Blk{p, n});
    return p;
}
struct Scope {  // one driver call: everything the translated code allocates inside it dies with it (no destructors: the
    Log log;    // translated types own nothing but such memory; message strings of errors are the exception and are small)
    Log* prev;
    Scope() : prev(cur()) { cur() = &log; }
    ~Scope() {
        cur() = prev;
        Cache& c = Cache::get();
        std::lock_guard<std::mutex> g(c.mu);
        for (const Blk& b : log.blocks) {
            if (b.n >= BIG && c.held + b.n <= CACHE_MAX) { c.free_.emplace(b.n, b.p); c.held += b.n; }
            else free(b.p);
        }
    }
};
struct Permanent {  // package-level state built lazily inside a call (go_init, sync.Once of package variables)
    Log* prev;
    Permanent() : prev(cur()) { cur() = nullptr; }
    ~Permanent() { cur() = prev; }
};
struct Tag {};
constexpr Tag tag{};
}  // namespace rt
}  // namespace go
inline void* operator new(size_t n, go::rt::Tag) { return go::rt::alloc(n); }
inline void operator delete(void*, go::rt::Tag) noexcept {}  // (a constructor that panics: the block stays in the call's log)
namespace go {

struct Panic { std::string msg; };
[[noreturn]] inline void panic_str(const std::string& m) { throw Panic{m}; }

struct Nil {
    template <class T> constexpr operator T*() const { return nullptr; }
    template <class F> operator std::function<F>() const { return std::function<F>(); }
};
constexpr Nil nil{};
template <class T> constexpr bool operator==(T* p, Nil) { return p == nullptr; }
template <class T> constexpr bool operator!=(T* p, Nil) { return p != nullptr; }
template <class T> constexpr bool operator==(Nil, T* p) { return p == nullptr; }
template <class T> constexpr bool operator!=(Nil, T* p) { return p != nullptr; }
template <class F> bool operator==(const std::function<F>& f, Nil) { return !f; }
template <class F> bool operator!=(const std::function<F>& f, Nil) { return (bool)f; }

typedef __int128 i128;

// ---- untyped integer constant ----
struct K {
    i128 v;
    constexpr K() : v(0) {}
    template <class T, class = typename std::enable_if<std::is_integral<T>::value && !std::is_same<T, bool>::value>::type>
    constexpr K(T x) : v((i128)x) {}
    static constexpr K raw(i128 x) { K k; k.v = x; return k; }
    constexpr operator double() const { return (double)v; }  // an untyped integer constant in a floating-point expression
    constexpr explicit operator bool() const = delete;
    friend constexpr K operator+(K a, K b) { return raw(a.v + b.v); }
    friend constexpr K operator-(K a, K b) { return raw(a.v - b.v); }
    friend constexpr K operator*(K a, K b) { return raw(a.v * b.v); }
    friend constexpr K operator/(K a, K b) { return raw(a.v / b.v); }
    friend constexpr K operator%(K a, K b) { return raw(a.v % b.v); }
    friend constexpr K operator&(K a, K b) { return raw(a.v & b.v); }
    friend constexpr K operator|(K a, K b) { return raw(a.v | b.v); }
    friend constexpr K operator^(K a, K b) { return raw(a.v ^ b.v); }
    friend constexpr K operator<<(K a, K b) { return raw(b.v >= 127 ? 0 : (i128)((unsigned __int128)a.v << (int)b.v)); }
    friend constexpr K operator>>(K a, K b) { return raw(b.v >= 127 ? (a.v < 0 ? -1 : 0) : a.v >> (int)b.v); }
    constexpr K operator-() const { return raw(-v); }
    constexpr K operator~() const { return raw(~v); }
    constexpr K operator+() const { return *this; }
    friend constexpr bool operator==(K a, K b) { return a.v == b.v; }
    friend constexpr bool operator!=(K a, K b) { return a.v != b.v; }
    friend constexpr bool operator<(K a, K b) { return a.v < b.v; }
    friend constexpr bool operator<=(K a, K b) { return a.v <= b.v; }
    friend constexpr bool operator>(K a, K b) { return a.v > b.v; }
    friend constexpr bool operator>=(K a, K b) { return a.v >= b.v; }
};
constexpr long long cn(K k) { return (long long)k.v; }  // array lengths and the like

// ---- typed integers ----
template <class T> struct I {
    typedef T raw_type;
    typedef typename std::make_unsigned<T>::type UT;
    static constexpr int W = 8 * (int)sizeof(T);
    T v;
    I() = default;  // (trivial: `I x{}` zero-initialises; huge tables of I are then zeroed with one memset, not element by element)
    constexpr I(K k) : v((T)k.v) {}
    template <class U> explicit constexpr I(I<U> o) : v((T)o.v) {}
    explicit constexpr I(double d) : v((T)d) {}
    explicit constexpr I(float d) : v((T)d) {}
    static constexpr I raw(T x) { I r{}; r.v = x; return r; }
    constexpr I* operator->() { return this; }
    friend constexpr I operator+(I a, I b) { return raw((T)((UT)a.v + (UT)b.v)); }
    friend constexpr I operator-(I a, I b) { return raw((T)((UT)a.v - (UT)b.v)); }
    friend constexpr I operator*(I a, I b) { return raw((T)((UT)a.v * (UT)b.v)); }
    friend I operator/(I a, I b) { if (b.v == 0) panic_str("integer divide by zero"); if (std::is_signed<T>::value && b.v == (T)-1) return raw((T)(0 - (UT)a.v)); return raw((T)(a.v / b.v)); }
    friend I operator%(I a, I b) { if (b.v == 0) panic_str("integer divide by zero"); if (std::is_signed<T>::value && b.v == (T)-1) return raw(0); return raw((T)(a.v % b.v)); }
    friend constexpr I operator&(I a, I b) { return raw((T)(a.v & b.v)); }
    friend constexpr I operator|(I a, I b) { return raw((T)(a.v | b.v)); }
    friend constexpr I operator^(I a, I b) { return raw((T)(a.v ^ b.v)); }
    constexpr I operator-() const { return raw((T)(0 - (UT)v)); }
    constexpr I operator~() const { return raw((T)~v); }
    constexpr I operator+() const { return *this; }
    static constexpr I shl(I a, unsigned long long n) { return n >= (unsigned)W ? raw(0) : raw((T)((UT)a.v << n)); }
    static constexpr I shr(I a, unsigned long long n) { return n >= (unsigned)W ? raw((T)(a.v < 0 ? -1 : 0)) : raw((T)(a.v >> n)); }
    template <class U> friend constexpr I operator<<(I a, I<U> n) { if (std::is_signed<U>::value && n.v < 0) panic_str("negative shift amount"); return shl(a, (unsigned long long)n.v); }
    template <class U> friend constexpr I operator>>(I a, I<U> n) { if (std::is_signed<U>::value && n.v < 0) panic_str("negative shift amount"); return shr(a, (unsigned long long)n.v); }
    friend constexpr I operator<<(I a, K n) { return shl(a, (unsigned long long)n.v); }
    friend constexpr I operator>>(I a, K n) { return shr(a, (unsigned long long)n.v); }
    friend constexpr bool operator==(I a, I b) { return a.v == b.v; }
    friend constexpr bool operator!=(I a, I b) { return a.v != b.v; }
    friend constexpr bool operator<(I a, I b) { return a.v < b.v; }
    friend constexpr bool operator<=(I a, I b) { return a.v <= b.v; }
    friend constexpr bool operator>(I a, I b) { return a.v > b.v; }
    friend constexpr bool operator>=(I a, I b) { return a.v >= b.v; }
    I& operator+=(I b) { *this = *this + b; return *this; }
    I& operator-=(I b) { *this = *this - b; return *this; }
    I& operator*=(I b) { *this = *this * b; return *this; }
    I& operator/=(I b) { *this = *this / b; return *this; }
    I& operator%=(I b) { *this = *this % b; return *this; }
    I& operator&=(I b) { *this = *this & b; return *this; }
    I& operator|=(I b) { *this = *this | b; return *this; }
    I& operator^=(I b) { *this = *this ^ b; return *this; }
    template <class S> I& operator<<=(S n) { *this = *this << n; return *this; }
    template <class S> I& operator>>=(S n) { *this = *this >> n; return *this; }
    I& operator++() { *this = *this + raw(1); return *this; }
    I& operator--() { *this = *this - raw(1); return *this; }
    I operator++(int) { I o = *this; ++*this; return o; }
    I operator--(int) { I o = *this; --*this; return o; }
};
// an untyped constant shifted by a typed count stays untyped (it takes its type from the context: the conversion truncates)
template <class U> constexpr K operator<<(K a, I<U> n) { return a << K((long long)n.v); }
template <class U> constexpr K operator>>(K a, I<U> n) { return a >> K((long long)n.v); }

typedef I<int64_t> Int;
typedef I<uint64_t> Uint;
typedef I<int8_t> int8;
typedef I<int16_t> int16;
typedef I<int32_t> int32;
typedef I<int64_t> int64;
typedef I<uint8_t> uint8;
typedef I<uint16_t> uint16;
typedef I<uint32_t> uint32;
typedef I<uint64_t> uint64;
typedef I<uint64_t> uintptr;
typedef I<uint8_t> byte;
typedef I<int32_t> rune;
typedef double float64;
typedef float float32;

template <class T> struct is_goint : std::false_type {};
template <class T> struct is_goint<I<T>> : std::true_type {};
template <class T> constexpr long long idx(I<T> i) { return (long long)i.v; }
constexpr long long idx(K k) { return (long long)k.v; }
template <class T, class = typename std::enable_if<std::is_base_of<I<typename T::raw_type>, T>::value && !is_goint<T>::value>::type>
constexpr long long idx(T i) { return (long long)i.v; }
inline double f64(K k) { return (double)k.v; }
template <class T> double f64(I<T> i) { return (double)i.v; }
inline double f64(double d) { return d; }

// x := <expr>: an untyped constant becomes int (float64), everything else keeps its type (by value)
inline Int def(K k) { return Int(k); }
inline double def(double d) { return d; }
template <class T> typename std::decay<T>::type def(T&& x) { return std::forward<T>(x); }

// ---- strings (byte strings; the translated code uses them for messages and []byte conversions only) ----
struct String {
    std::string s;
    String() {}
    String(const char* c) : s(c) {}
    String(const std::string& c) : s(c) {}
    String* operator->() { return this; }
    friend String operator+(const String& a, const String& b) { return String(a.s + b.s); }
    friend bool operator==(const String& a, const String& b) { return a.s == b.s; }
    friend bool operator!=(const String& a, const String& b) { return a.s != b.s; }
    template <class X> byte operator[](X i) const { long long k = idx(i); if (k < 0 || k >= (long long)s.size()) panic_str("string index out of range"); return byte::raw((uint8_t)s[(size_t)k]); }
};

// ---- slices and arrays ----
[[noreturn]] inline void oob(const char* what, long long i, long long n) {
    char b[128];
    snprintf(b, sizeof b, "runtime error: %s out of range [%lld] with length/capacity %lld", what, i, n);
    panic_str(b);
}
template <class T> struct Slice {
    T* p;
    long long n, c;
    Slice() : p(nullptr), n(0), c(0) {}
    Slice(Nil) : p(nullptr), n(0), c(0) {}
    Slice(T* p_, long long n_, long long c_) : p(p_), n(n_), c(c_) {}
    Slice(std::initializer_list<T> il) : p(nullptr), n((long long)il.size()), c((long long)il.size()) {
        p = (T*)rt::alloc((il.size() ? il.size() : 1) * sizeof(T));
        long long k = 0;
        for (const T& x : il) new (&p[k++]) T(x);
    }
    Slice* operator->() { return this; }
    template <class X> T& operator[](X i) const { const long long k = idx(i); if (k < 0 || k >= n) oob("index", k, n); return p[k]; }
    Slice sl(long long lo, long long hi) const { if (hi < 0 || hi > c) oob("slice bounds", hi, c); if (lo < 0 || lo > hi) oob("slice bounds", lo, hi); return Slice(p + lo, hi - lo, c - lo); }
    Slice sl3(long long lo, long long hi, long long mx) const { if (mx < 0 || mx > c) oob("slice bounds", mx, c); if (hi < 0 || hi > mx) oob("slice bounds", hi, mx); if (lo < 0 || lo > hi) oob("slice bounds", lo, hi); return Slice(p + lo, hi - lo, mx - lo); }
    friend bool operator==(const Slice& a, Nil) { return a.p == nullptr; }
    friend bool operator!=(const Slice& a, Nil) { return a.p != nullptr; }
};
template <class T, long long N> struct Array {
    T a[N > 0 ? N : 1];
    Array() { if (std::is_trivially_default_constructible<T>::value) memset((void*)a, 0, sizeof a); }  // (not constexpr on purpose: the
    // compiler must not try to constant-evaluate the zero value of a 4-million-entry table)
    Array* operator->() { return this; }
    template <class X> T& operator[](X i) { const long long k = idx(i); if (k < 0 || k >= N) oob("index", k, N); return a[k]; }
    template <class X> const T& operator[](X i) const { const long long k = idx(i); if (k < 0 || k >= N) oob("index", k, N); return a[k]; }
    Slice<T> sl(long long lo, long long hi) { if (hi < 0 || hi > N) oob("slice bounds", hi, N); if (lo < 0 || lo > hi) oob("slice bounds", lo, hi); return Slice<T>(a + lo, hi - lo, N - lo); }
    Slice<T> sl3(long long lo, long long hi, long long mx) { return Slice<T>(a, N, N).sl3(lo, hi, mx); }
    friend bool operator==(const Array& x, const Array& y) { for (long long i = 0; i < N; i++) if (!(x.a[i] == y.a[i])) return false; return true; }
    friend bool operator!=(const Array& x, const Array& y) { return !(x == y); }
};
// slicing x[lo:hi], x[lo:hi:max] with omitted bounds (-1): slices, arrays, pointers to arrays, strings
template <class T> long long len_(const Slice<T>& s) { return s.n; }
template <class T> long long cap_(const Slice<T>& s) { return s.c; }
template <class T, long long N> long long len_(const Array<T, N>&) { return N; }
template <class T, long long N> long long cap_(const Array<T, N>&) { return N; }
template <class T, long long N> long long len_(Array<T, N>* const&) { return N; }
template <class T, long long N> long long cap_(Array<T, N>* const&) { return N; }
inline long long len_(const String& s) { return (long long)s.s.size(); }
template <class S> Int len(const S& s) { return Int::raw(len_(s)); }
template <class S> Int cap(const S& s) { return Int::raw(cap_(s)); }
template <class T> Slice<T> slice(const Slice<T>& s, long long lo, long long hi) { return s.sl(lo < 0 ? 0 : lo, hi < 0 ? s.n : hi); }
template <class T> Slice<T> slice3(const Slice<T>& s, long long lo, long long hi, long long mx) { return s.sl3(lo < 0 ? 0 : lo, hi, mx); }
template <class T, long long N> Slice<T> slice(Array<T, N>& s, long long lo, long long hi) { return s.sl(lo < 0 ? 0 : lo, hi < 0 ? N : hi); }
template <class T, long long N> Slice<T> slice(Array<T, N>* s, long long lo, long long hi) { return s->sl(lo < 0 ? 0 : lo, hi < 0 ? N : hi); }
template <class T, long long N> Slice<T> slice3(Array<T, N>& s, long long lo, long long hi, long long mx) { return s.sl3(lo < 0 ? 0 : lo, hi, mx); }
inline String slice(const String& s, long long lo, long long hi) { if (lo < 0) lo = 0; if (hi < 0) hi = (long long)s.s.size(); if (hi > (long long)s.s.size() || lo > hi) oob("slice bounds", hi, (long long)s.s.size()); return String(s.s.substr((size_t)lo, (size_t)(hi - lo))); }
// indexing through a pointer to an array (Go dereferences it)
template <class T, long long N, class X> T& at(Array<T, N>* a, X i) { return (*a)[i]; }
template <class C, class X> auto at(C&& c, X i) -> decltype(c[i]) { return c[i]; }

template <class T> Slice<T> make_slice(long long n, long long c = -1) {
    if (c < 0) c = n;
    if (n < 0 || c < n) panic_str("makeslice: len out of range");
    T* p = (T*)rt::alloc((size_t)(c ? c : 1) * sizeof(T));
    if (!std::is_trivially_default_constructible<T>::value || true) for (long long i = 0; i < c; i++) new (&p[i]) T();
    return Slice<T>(p, n, c);
}
template <class T> Slice<T> grow_(Slice<T> s, long long need) {
    if (need <= s.c) return s;
    long long nc = s.c < 256 ? 2 * s.c : s.c + s.c / 4 + 192;
    if (nc < need) nc = need;
    T* p = (T*)rt::alloc((size_t)(nc ? nc : 1) * sizeof(T));
    for (long long i = 0; i < nc; i++) new (&p[i]) T();
    for (long long i = 0; i < s.n; i++) p[i] = s.p[i];
    return Slice<T>(p, s.n, nc);
}
template <class T> Slice<T> append(Slice<T> s) { return s; }
template <class T, class... R> Slice<T> append(Slice<T> s, const T& x, const R&... rest) {
    Slice<T> g = grow_(s, s.n + 1 + (long long)sizeof...(rest));
    g.p[g.n++] = x;
    return append(g, rest...);
}
template <class T, class... R> Slice<T> append(Slice<T> s, K x, const R&... rest) { return append(s, T(x), rest...); }
template <class T> Slice<T> append(Nil, const T& x) { return append(Slice<T>(), x); }
template <class T> Slice<T> append_all(Slice<T> s, const Slice<T>& t) {
    Slice<T> g = grow_(s, s.n + t.n);
    memmove((void*)(g.p + g.n), (const void*)t.p, (size_t)t.n * sizeof(T));  // (may overlap: append(x[:a], x[b:]...))
    g.n += t.n;
    return g;
}
inline Slice<byte> append_all(Slice<byte> s, const String& t) {
    Slice<byte> g = grow_(s, s.n + (long long)t.s.size());
    memcpy((void*)(g.p + g.n), t.s.data(), t.s.size());
    g.n += (long long)t.s.size();
    return g;
}
template <class T> Int copy(Slice<T> d, const Slice<T>& s) {
    const long long k = d.n < s.n ? d.n : s.n;
    if (k > 0) memmove((void*)d.p, (const void*)s.p, (size_t)k * sizeof(T));
    return Int::raw(k);
}
inline Int copy(Slice<byte> d, const String& s) {
    const long long k = d.n < (long long)s.s.size() ? d.n : (long long)s.s.size();
    if (k > 0) memcpy((void*)d.p, s.s.data(), (size_t)k);
    return Int::raw(k);
}
inline Slice<byte> to_bytes(const String& s) { Slice<byte> r = make_slice<byte>((long long)s.s.size()); if (!s.s.empty()) memcpy((void*)r.p, s.s.data(), s.s.size()); return r; }
inline Slice<byte> to_bytes(const Slice<byte>& s) { return s; }
inline String to_string(const Slice<byte>& b) { return String(std::string((const char*)b.p, (size_t)b.n)); }
inline String to_string(const String& s) { return s; }
template <class T> T* new_() { return new (rt::tag) T(); }
template <class T, long long N> Array<T, N>& ix(Array<T, N>* p) { return *p; }
template <class C> C&& ix(C&& c) { return std::forward<C>(c); }
template <class A, class T> A* as_array(const Slice<T>& s) { if (s.n < (long long)(sizeof(A) / sizeof(T))) panic_str("cannot convert slice to array pointer: too short"); return reinterpret_cast<A*>(s.p); }
template <class B, class D> B* base(D* d) { return static_cast<B*>(d); }
template <class B, class D> B* base(D& d) { return static_cast<B*>(&d); }
struct Any { Any() {} template <class T> Any(const T&) {} Any* operator->() { return this; } };
// x.(*T) on a translated interface value: the adapter remembers the concrete type it was made from
template <class T> inline const void* type_tag() { static const char tag = 0; return &tag; }
template <class P, class Iface> inline std::tuple<P, bool> cast(const Iface& i) {
    typedef typename std::remove_pointer<P>::type T;
    if (i.b_ != nullptr && i.b_->tid_() == type_tag<T>()) return std::tuple<P, bool>((P)i.b_->obj_(), true);
    return std::tuple<P, bool>((P) nullptr, false);
}
// recover(): a panic of the translated code is a C++ exception that travels to the driver; there is never anything to recover
struct Recovered { friend bool operator==(const Recovered&, Nil) { return true; } friend bool operator!=(const Recovered&, Nil) { return false; } };
inline Recovered recover_() { return Recovered{}; }
inline Recovered def(Recovered r) { return r; }

template <class T> void clear(const Slice<T>& s) { for (long long i = 0; i < s.n; i++) s.p[i] = T(); }
// min / max builtins (Go 1.21)
template <class T> T min(T a, T b) { return b < a ? b : a; }
template <class T> T max(T a, T b) { return a < b ? b : a; }
template <class T> I<T> min(I<T> a, K b) { return min(a, I<T>(b)); }
template <class T> I<T> max(I<T> a, K b) { return max(a, I<T>(b)); }
template <class T> I<T> min(K a, I<T> b) { return min(I<T>(a), b); }
template <class T> I<T> max(K a, I<T> b) { return max(I<T>(a), b); }

// ---- errors, panics, defer ----
struct ErrorObj { std::string msg; };
struct error {
    const ErrorObj* e;
    error() : e(nullptr) {}
    error(Nil) : e(nullptr) {}
    explicit error(const ErrorObj* o) : e(o) {}
    error* operator->() { return this; }
    String Error() const { return String(e ? e->msg : "<nil>"); }
    friend bool operator==(const error& a, const error& b) { return a.e == b.e; }
    friend bool operator!=(const error& a, const error& b) { return a.e != b.e; }
    friend bool operator==(const error& a, Nil) { return a.e == nullptr; }
    friend bool operator!=(const error& a, Nil) { return a.e != nullptr; }
};
[[noreturn]] inline void panic(const error& e) { panic_str(e.e ? e.e->msg : "panic(nil error)"); }
[[noreturn]] inline void panic(const String& s) { panic_str(s.s); }
[[noreturn]] inline void panic(const char* s) { panic_str(s); }
template <class T> [[noreturn]] void panic(const T&) { panic_str("panic"); }
struct Defer {
    std::function<void()> f;
    template <class F> explicit Defer(F&& g) : f(std::forward<F>(g)) {}
    ~Defer() noexcept(false) { f(); }
};
template <class... A> void println(const A&...) {}
template <class... A> void print(const A&...) {}

template <class T> bool is_nil(T* p) { return p == nullptr; }

}  // namespace go

// ===================== standard-library pieces the translated sources call =====================
namespace errors {
inline go::error New(const go::String& s) { return go::error(new go::ErrorObj{s.s}); }
inline bool Is(const go::error& a, const go::error& b) { return a == b; }
}  // namespace errors
namespace fmt {
template <class... A> go::error Errorf(const go::String& f, const A&...) { return go::error(new go::ErrorObj{f.s}); }
template <class... A> go::String Sprintf(const go::String& f, const A&...) { return f; }
template <class... A> go::String Sprint(const A&...) { return go::String(""); }
template <class... A> go::String Sprintln(const A&...) { return go::String(""); }
template <class... A> void Println(const A&...) {}
template <class... A> void Printf(const A&...) {}
template <class... A> void Print(const A&...) {}
}  // namespace fmt
namespace strconv {
static constexpr go::K IntSize = go::K(64LL);  // strconv.IntSize on a 64-bit platform
}
namespace bits {
using namespace go;
inline Int Len8(uint8 x) { return Int::raw(x.v ? 32 - __builtin_clz((unsigned)x.v) : 0); }
inline Int Len16(uint16 x) { return Int::raw(x.v ? 32 - __builtin_clz((unsigned)x.v) : 0); }
inline Int Len32(uint32 x) { return Int::raw(x.v ? 32 - __builtin_clz(x.v) : 0); }
inline Int Len64(uint64 x) { return Int::raw(x.v ? 64 - __builtin_clzll(x.v) : 0); }
inline Int Len(Uint x) { return Len64(x); }
inline Int TrailingZeros8(uint8 x) { return Int::raw(x.v ? __builtin_ctz((unsigned)x.v) : 8); }
inline Int TrailingZeros16(uint16 x) { return Int::raw(x.v ? __builtin_ctz((unsigned)x.v) : 16); }
inline Int TrailingZeros32(uint32 x) { return Int::raw(x.v ? __builtin_ctz(x.v) : 32); }
inline Int TrailingZeros64(uint64 x) { return Int::raw(x.v ? __builtin_ctzll(x.v) : 64); }
inline Int LeadingZeros32(uint32 x) { return Int::raw(x.v ? __builtin_clz(x.v) : 32); }
inline Int LeadingZeros64(uint64 x) { return Int::raw(x.v ? __builtin_clzll(x.v) : 64); }
inline Int OnesCount32(uint32 x) { return Int::raw(__builtin_popcount(x.v)); }
inline Int OnesCount64(uint64 x) { return Int::raw(__builtin_popcountll(x.v)); }
constexpr go::K UintSize(64);
template <class S> uint32 RotateLeft32(uint32 x, S k) { const int s = (int)(go::idx(k) & 31); return uint32::raw((x.v << s) | (x.v >> ((32 - s) & 31))); }
template <class S> uint64 RotateLeft64(uint64 x, S k) { const int s = (int)(go::idx(k) & 63); return uint64::raw((x.v << s) | (x.v >> ((64 - s) & 63))); }
inline uint32 ReverseBytes32(uint32 x) { return uint32::raw(__builtin_bswap32(x.v)); }
inline uint64 ReverseBytes64(uint64 x) { return uint64::raw(__builtin_bswap64(x.v)); }
}  // namespace bits
namespace binary {
using namespace go;
struct LittleEndian_t {
    LittleEndian_t* operator->() { return this; }
    uint16 Uint16(const Slice<byte>& b) const { (void)b[K(1)]; uint16_t v; memcpy(&v, b.p, 2); return uint16::raw(v); }
    uint32 Uint32(const Slice<byte>& b) const { (void)b[K(3)]; uint32_t v; memcpy(&v, b.p, 4); return uint32::raw(v); }
    uint64 Uint64(const Slice<byte>& b) const { (void)b[K(7)]; uint64_t v; memcpy(&v, b.p, 8); return uint64::raw(v); }
    void PutUint16(const Slice<byte>& b, uint16 x) const { (void)b[K(1)]; memcpy(b.p, &x.v, 2); }
    void PutUint32(const Slice<byte>& b, uint32 x) const { (void)b[K(3)]; memcpy(b.p, &x.v, 4); }
    void PutUint64(const Slice<byte>& b, uint64 x) const { (void)b[K(7)]; memcpy(b.p, &x.v, 8); }
    Slice<byte> AppendUint32(Slice<byte> b, uint32 x) const { return go::append(b, byte(x), byte(x >> K(8)), byte(x >> K(16)), byte(x >> K(24))); }
};
static LittleEndian_t LittleEndian;
constexpr K MaxVarintLen16(3), MaxVarintLen32(5), MaxVarintLen64(10);
inline Int PutUvarint(const Slice<byte>& buf, uint64 x) {
    long long i = 0;
    uint64_t v = x.v;
    while (v >= 0x80) { buf[K(i)] = byte::raw((uint8_t)(v | 0x80)); v >>= 7; i++; }
    buf[K(i)] = byte::raw((uint8_t)v);
    return Int::raw(i + 1);
}
// encoding/binary.PutVarint: zig-zag, then PutUvarint
inline Int PutVarint(const Slice<byte>& buf, int64 x) {
    uint64_t ux = (uint64_t)x.v << 1;
    if (x.v < 0) ux = ~ux;
    return PutUvarint(buf, uint64::raw(ux));
}
inline std::tuple<uint64, Int> Uvarint(const Slice<byte>& buf) {
    uint64_t x = 0;
    unsigned sft = 0;
    for (long long i = 0; i < buf.n; i++) {
        const uint8_t b = buf.p[i].v;
        if (i == 10) return {uint64::raw(0), Int::raw(-(i + 1))};
        if (b < 0x80) {
            if (i == 9 && b > 1) return {uint64::raw(0), Int::raw(-(i + 1))};
            return {uint64::raw(x | ((uint64_t)b << sft)), Int::raw(i + 1)};
        }
        x |= (uint64_t)(b & 0x7f) << sft;
        sft += 7;
    }
    return {uint64::raw(0), Int::raw(0)};
}
struct BigEndian_t {
    BigEndian_t* operator->() { return this; }
    uint32 Uint32(const Slice<byte>& b) const { (void)b[K(3)]; uint32_t v; memcpy(&v, b.p, 4); return uint32::raw(__builtin_bswap32(v)); }
    uint64 Uint64(const Slice<byte>& b) const { (void)b[K(7)]; uint64_t v; memcpy(&v, b.p, 8); return uint64::raw(__builtin_bswap64(v)); }
    void PutUint32(const Slice<byte>& b, uint32 x) const { (void)b[K(3)]; uint32_t v = __builtin_bswap32(x.v); memcpy(b.p, &v, 4); }
    void PutUint64(const Slice<byte>& b, uint64 x) const { (void)b[K(7)]; uint64_t v = __builtin_bswap64(x.v); memcpy(b.p, &v, 8); }
};
static BigEndian_t BigEndian;
}  // namespace binary
namespace bytes {
inline bool Equal(const go::Slice<go::byte>& a, const go::Slice<go::byte>& b) { return a.n == b.n && (a.n == 0 || memcmp(a.p, b.p, (size_t)a.n) == 0); }
}  // namespace bytes
namespace math {
constexpr go::K MaxUint8(255), MaxUint16(65535), MaxInt16(32767), MaxInt32(2147483647LL), MaxUint32(4294967295LL), MaxInt64(9223372036854775807LL),
    MaxUint64(18446744073709551615ULL), MaxInt8(127), MinInt32(-2147483648LL), MaxInt(9223372036854775807LL);
// math.Log2 as the Go standard library computes it (src/math/log10.go: frexp, then log2(frac) = Log(frac) * (1/Ln2), exact for
// powers of two; src/math/log.go: the FDLIBM e_log.c algorithm in float64 operations, no fused multiply-add on amd64)
inline double log_go(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                 L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (std::isnan(x) || std::isinf(x) && x > 0) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < 0.70710678118654752440 /* Sqrt2/2 */) { f1 *= 2; ki--; }
    volatile double f = f1 - 1;
    const double k = (double)ki;
    volatile double s = f / (2 + f);
    volatile double s2 = s * s;
    volatile double s4 = s2 * s2;
    volatile double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    volatile double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    volatile double R = t1 + t2;
    volatile double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
inline double Log(double x) { return log_go(x); }
inline double Log2(double x) {
    int e;
    const double frac = frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);  // exact for powers of two
    const double Ln2 = 0.693147180559945309417232121458176568;
    volatile double l = log_go(frac) * (1 / Ln2);
    return l + (double)e;
}
inline double Pow(double x, double y) { return pow(x, y); }
inline double Ceil(double x) { return ceil(x); }
inline double Floor(double x) { return floor(x); }
inline double Sqrt(double x) { return sqrt(x); }
inline double Abs(double x) { return fabs(x); }
inline double Float64frombits(go::uint64 b) { double d; memcpy(&d, &b.v, 8); return d; }
inline go::uint64 Float64bits(double d) { uint64_t b; memcpy(&b, &d, 8); return go::uint64::raw(b); }
}  // namespace math
namespace cpuinfo {  // internal/cpuinfo: what the amd64 flavour's dispatch helpers ask (the driver can force the non-BMI2 routines)
inline int& force() { static int f = -1; return f; }  // -1: the host's CPU; 0: pretend BMI1/BMI2 are absent
inline bool HasBMI2() { return force() != 0 && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi"); }
inline bool HasBMI1() { return force() != 0 && __builtin_cpu_supports("bmi"); }
}  // namespace cpuinfo
namespace sync {
struct Once {
    bool done = false;
    Once* operator->() { return this; }
    template <class F> void Do(F&& f) { if (!done) { done = true; f(); } }
};
struct Mutex { Mutex* operator->() { return this; } void Lock() {} void Unlock() {} };
struct WaitGroup { WaitGroup* operator->() { return this; } void Add(go::K) {} void Done() {} void Wait() {} };
struct Pool { Pool* operator->() { return this; } template <class T> void Put(const T&) {} };
}  // namespace sync
namespace io {
// io.Reader / io.Writer: the driver's source / sink implement ReaderImpl / WriterImpl (what `r` of s2.NewReader(r) and `w` of
// zstd.NewWriter(w, ...) are to the reference).  A nil Reader reads as an endless source of zeroes (rand_::Reader: the padding source)
struct ReaderImpl { virtual std::tuple<go::Int, go::error> Read(go::Slice<go::byte> p) = 0; virtual ~ReaderImpl() {} };
struct Reader {
    ReaderImpl* p = nullptr;
    Reader() {}
    Reader(ReaderImpl* q) : p(q) {}
    Reader(go::Nil) {}
    Reader* operator->() { return this; }
    std::tuple<go::Int, go::error> Read(go::Slice<go::byte> b) const { if (!p) return {go::len(b), go::error()}; return p->Read(b); }
    friend bool operator==(const Reader& a, go::Nil) { return a.p == nullptr; }
    friend bool operator!=(const Reader& a, go::Nil) { return a.p != nullptr; }
};
struct WriterImpl { virtual std::tuple<go::Int, go::error> Write(go::Slice<go::byte> p) = 0; virtual ~WriterImpl() {} };
struct Writer {
    WriterImpl* p = nullptr;
    Writer() {}
    Writer(WriterImpl* q) : p(q) {}
    Writer(go::Nil) {}
    WriterImpl* operator->() const { if (!p) go::panic_str("nil io.Writer"); return p; }
    friend bool operator==(const Writer& a, go::Nil) { return a.p == nullptr; }
    friend bool operator!=(const Writer& a, go::Nil) { return a.p != nullptr; }
};
static const go::error ErrUnexpectedEOF = go::error(new go::ErrorObj{"unexpected EOF"});
static const go::error EOF_ = go::error(new go::ErrorObj{"EOF"});
static const go::error ErrShortBuffer = go::error(new go::ErrorObj{"short buffer"});
static const go::error ErrShortWrite = go::error(new go::ErrorObj{"short write"});
// io.ReadFull (io.ReadAtLeast with min = len(buf)): EOF only if no byte was read, ErrUnexpectedEOF after a partial read
inline std::tuple<go::Int, go::error> ReadFull(const Reader& r, const go::Slice<go::byte>& b) {
    long long n = 0;
    go::error err;
    while (n < b.n && err == go::nil) {
        auto t = r.Read(b.sl(n, b.n));
        n += std::get<0>(t).v;
        err = std::get<1>(t);
    }
    if (n >= b.n) err = go::error();
    else if (n > 0 && err == EOF_) err = ErrUnexpectedEOF;
    return {go::Int::raw(n), err};
}
}  // namespace io
namespace crc32 {  // hash/crc32 of the Go standard library: table-driven, reflected (s2 uses the Castagnoli polynomial)
struct Table { uint32_t t[256]; };
static constexpr go::K Castagnoli = go::K(0x82f63b78LL);
inline Table* MakeTable(go::K poly) {
    Table* tb = new Table();
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ (uint32_t)poly.v : c >> 1;
        tb->t[i] = c;
    }
    return tb;
}
inline go::uint32 Update(go::uint32 crc, Table* tab, const go::Slice<go::byte>& p) {
    uint32_t c = ~crc.v;
    for (long long i = 0; i < p.n; i++) c = tab->t[(c ^ p.p[i].v) & 0xFF] ^ (c >> 8);
    return go::uint32::raw(~c);
}
}  // namespace crc32
namespace rdebug {
inline void PrintStack() {}
}
namespace rand_ {
static io::Reader Reader;
}
namespace slices {
template <class T> T Max(const go::Slice<T>& s) { if (s.n == 0) go::panic_str("slices.Max: empty list"); T m = s.p[0]; for (long long i = 1; i < s.n; i++) if (m < s.p[i]) m = s.p[i]; return m; }
}
namespace log_ {
template <class... A> void Println(const A&...) {}
template <class... A> void Printf(const A&...) {}
template <class... A> void Print(const A&...) {}
}  // namespace log_
namespace runtime {
inline go::Int GOMAXPROCS(go::K) { return go::Int(go::K(1)); }
}
namespace race {
template <class... A> void ReadSlice(const A&...) {}
template <class... A> void WriteSlice(const A&...) {}
}  // namespace race
// internal/le (the reference's own five-line load/store helpers over encoding/binary: unsafe_disabled.go)
namespace le {
using namespace go;
template <class X> byte Load8(const Slice<byte>& b, X i) { return b[i]; }
template <class X> uint16 Load16(const Slice<byte>& b, X i) { return binary::LittleEndian.Uint16(go::slice(b, go::idx(i), -1)); }
template <class X> uint32 Load32(const Slice<byte>& b, X i) { return binary::LittleEndian.Uint32(go::slice(b, go::idx(i), -1)); }
template <class X> uint64 Load64(const Slice<byte>& b, X i) { return binary::LittleEndian.Uint64(go::slice(b, go::idx(i), -1)); }
inline void Store16(const Slice<byte>& b, uint16 v) { binary::LittleEndian.PutUint16(b, v); }
inline void Store32(const Slice<byte>& b, uint32 v) { binary::LittleEndian.PutUint32(b, v); }
inline void Store64(const Slice<byte>& b, uint64 v) { binary::LittleEndian.PutUint64(b, v); }
}  // namespace le

namespace go {
// for ... range x
template <class T> long long rangelen(const Slice<T>& s) { return s.n; }
template <class T, long long N> long long rangelen(const Array<T, N>&) { return N; }
template <class T, long long N> long long rangelen(Array<T, N>* const&) { return N; }
inline long long rangelen(const String& s) { return (long long)s.s.size(); }
template <class T> long long rangelen(I<T> n) { return (long long)n.v; }
inline long long rangelen(K n) { return (long long)n.v; }
template <class T> T rangeval(const Slice<T>& s, long long i) { return s.p[i]; }
template <class T, long long N> T rangeval(const Array<T, N>& s, long long i) { return s.a[i]; }
template <class T, long long N> T rangeval(Array<T, N>* const& s, long long i) { return s->a[i]; }
}  // namespace go
