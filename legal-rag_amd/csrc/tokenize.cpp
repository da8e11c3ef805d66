// Query tokeniser + vocabulary lookup of the BM25 channel, native and batched (host code, no device work).
//
// Replaces, per query, `tokens = list(jieba.cut(query))` + the term lookup of rank_bm25's get_scores
// (legalrag/retrieval/bm25_retriever.py:73-74) for text WITHOUT Han characters — the case jieba's default mode
// (cut_all=False, HMM=True) decides without its dictionary; legal-rag_amd/text.py states the rule and is the
// executable specification this file follows token for token (tests/test_text.py compares the two):
//   1. the sentence is split on maximal runs of [一-鿕 a-zA-Z0-9 + # & . _ % -] ("blocks"); between blocks
//      "\r\n" or ONE whitespace character (Python's \s on str: str.isspace()) is a token, every other character is
//      a token of its own;
//   2. a block of one character is that character; a longer block is cut by finalseg's non-Han rule: runs matching
//      [a-zA-Z0-9]+(?:\.\d+)?%? are one token each and each maximal run of the remaining characters is one token;
//      the five ASCII multi-character entries of jieba's dictionary (AT&T, C++, c++, C#, c#) are tokens wherever
//      they start, the text between them is cut as above.
// A query that holds a Han character is NOT tokenised here (it needs jieba's dictionary): it is flagged and the
// caller takes the Python path (which raises unless a segmenter or the explicit stand-in is configured).
// Queries are NOT lower-cased (the reference does not, bm25_retriever.py:73).  One call handles a whole batch and
// writes the term-id CSR amdr_bm25_search takes; ctypes releases the GIL for its duration.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <pthread.h>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

// The vocabulary as an open-addressing table over ONE copy of the term bytes: a lookup hashes the token's bytes where
// they lie in the query (no std::string is built per token) and compares with memcmp.
struct amdr_tokenizer {
  std::string blob;               // all terms, back to back
  std::vector<int64_t> offs;      // term i = blob[offs[i] .. offs[i + 1])
  std::vector<int32_t> slots;     // -1 = empty, else a term id; size = power of two >= 2 x terms
  uint32_t mask = 0;
  int32_t single[256];            // one-byte tokens (blanks and punctuation are two thirds of a query's tokens): direct
  static inline uint32_t hash(const unsigned char* p, size_t n) {  // FNV-1a, folded
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return (uint32_t)(h ^ (h >> 32));
  }
  inline int32_t find(const unsigned char* p, size_t n) const {
    if (n == 1) return single[p[0]];
    return find_slow(p, n);
  }
  inline int32_t find_slow(const unsigned char* p, size_t n) const {
    if (slots.empty()) return -1;
    for (uint32_t i = hash(p, n) & mask;; i = (i + 1) & mask) {
      const int32_t id = slots[i];
      if (id < 0) return -1;
      const int64_t lo = offs[id];
      if ((size_t)(offs[id + 1] - lo) == n && memcmp(blob.data() + lo, p, n) == 0) return id;
    }
  }
};

namespace {

inline bool is_alnum(uint32_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
inline bool is_digit(uint32_t c) { return c >= '0' && c <= '9'; }
inline bool is_han(uint32_t c) { return c >= 0x4E00 && c <= 0x9FD5; }
inline bool is_block(uint32_t c) {
  return is_alnum(c) || is_han(c) || c == '+' || c == '#' || c == '&' || c == '.' || c == '_' || c == '%' || c == '-';
}
// Python str.isspace() == what \s matches in a str pattern
inline bool is_space(uint32_t c) {
  return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
         (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

// decode one UTF-8 code point at p (< end); malformed bytes are taken one at a time (Python str input cannot be
// malformed; this only keeps the scan inside the buffer)
inline uint32_t decode(const unsigned char* p, const unsigned char* end, int* len) {
  const unsigned char b = *p;
  if (b < 0x80) {
    *len = 1;
    return b;
  }
  int n = (b >= 0xF0) ? 4 : (b >= 0xE0) ? 3 : (b >= 0xC0) ? 2 : 1;
  if (n == 1 || p + n > end) {
    *len = 1;
    return 0xFFFD;
  }
  uint32_t c = b & (0xFF >> (n + 1));
  for (int i = 1; i < n; ++i) c = (c << 6) | (p[i] & 0x3F);
  *len = n;
  return c;
}

struct Span {
  int32_t lo, hi;  // byte range of a token
};

// finalseg's non-Han rule on an ASCII buffer [lo, hi)
template <class Emit>
inline void finalseg_ascii(const unsigned char* s, int lo, int hi, Emit&& emit) {
  int i = lo;
  while (i < hi) {
    int j = i;
    if (is_alnum(s[i])) {
      while (j < hi && is_alnum(s[j])) ++j;
      if (j + 1 < hi && s[j] == '.' && is_digit(s[j + 1])) {
        ++j;
        while (j < hi && is_digit(s[j])) ++j;
      }
      if (j < hi && s[j] == '%') ++j;
    } else {
      while (j < hi && !is_alnum(s[j])) ++j;
    }
    emit(i, j);
    i = j;
  }
}

inline int dict_word_at(const unsigned char* s, int i, int hi) {  // length of an ASCII dictionary word starting at i, or 0
  static const char* const kWords[] = {"AT&T", "C++", "c++", "C#", "c#"};
  for (const char* w : kWords) {
    const int n = (int)strlen(w);
    if (i + n <= hi && memcmp(s + i, w, n) == 0) return n;
  }
  return 0;
}

// a block without Han characters (ASCII by construction)
template <class Emit>
inline void cut_block(const unsigned char* s, int lo, int hi, Emit&& emit) {
  bool marks = false;
  for (int i = lo; i < hi; ++i) marks |= (s[i] == '&' || s[i] == '+' || s[i] == '#');
  auto flush = [&](int a, int b) {
    if (b - a == 1)
      emit(a, b);
    else if (b > a)
      finalseg_ascii(s, a, b, emit);
  };
  if (!marks) {
    flush(lo, hi);
    return;
  }
  int buf = lo, i = lo;
  while (i < hi) {
    const int n = dict_word_at(s, i, hi);
    if (n) {
      flush(buf, i);
      emit(i, i + n);
      i += n;
      buf = i;
    } else {
      ++i;
    }
  }
  flush(buf, hi);
}

// byte classes of the ASCII fast path: 1 = block character ([a-zA-Z0-9+#&._%-]), 0 = anything else
struct AsciiBlock {
  unsigned char t[256];
  AsciiBlock() {
    for (int c = 0; c < 256; ++c) t[c] = (c < 0x80 && is_block((uint32_t)c)) ? 1 : 0;
  }
};
static const AsciiBlock kAsciiBlock;

// tokens of one sentence; returns false (nothing emitted is meaningful) when it holds a Han character
template <class Emit>
inline bool tokenize(const unsigned char* s, int n, Emit&& emit) {
  // pure ASCII (every English query): no decoding, no Han check, one table lookup per byte — the same rule
  bool ascii = true;
  for (int i = 0; i < n; ++i) ascii &= s[i] < 0x80;
  if (ascii) {
    int i = 0;
    while (i < n) {
      if (kAsciiBlock.t[s[i]]) {
        int j = i + 1;
        while (j < n && kAsciiBlock.t[s[j]]) ++j;
        cut_block(s, i, j, emit);
        i = j;
      } else if (s[i] == '\r' && i + 1 < n && s[i + 1] == '\n') {
        emit(i, i + 2);
        i += 2;
      } else {
        emit(i, i + 1);
        ++i;
      }
    }
    return true;
  }
  const unsigned char* end = s + n;
  for (int i = 0; i < n;) {  // Han anywhere -> the whole sentence goes to the caller's segmenter
    int len;
    if (is_han(decode(s + i, end, &len))) return false;
    i += len;
  }
  int i = 0;
  while (i < n) {
    int len;
    const uint32_t c = decode(s + i, end, &len);
    if (is_block(c)) {
      int j = i;
      while (j < n) {
        int l2;
        if (!is_block(decode(s + j, end, &l2))) break;
        j += l2;
      }
      cut_block(s, i, j, emit);
      i = j;
    } else if (c == '\r' && i + 1 < n && s[i + 1] == '\n') {
      emit(i, i + 2);
      i += 2;
    } else {  // one whitespace character, or any other character on its own
      emit(i, i + len);
      i += len;
    }
  }
  return true;
}

}  // namespace

extern "C" {

int amdr_tokenizer_create(const char* vocab_blob, const int64_t* vocab_offsets, int64_t n_terms,
                          amdr_tokenizer_t** out) {
  AMDR_REQUIRE(out != nullptr, "tokenizer_create: out is null");
  *out = nullptr;
  AMDR_REQUIRE(n_terms >= 0 && n_terms < (1ll << 30) && (n_terms == 0 || (vocab_blob && vocab_offsets)),
               "tokenizer_create: bad vocabulary");
  amdr_tokenizer* t = new (std::nothrow) amdr_tokenizer();
  if (!t) return amdr::fail(AMDR_ENOMEM, "tokenizer_create: host alloc");
  for (int c = 0; c < 256; ++c) t->single[c] = -1;
  for (int64_t i = 0; i < n_terms; ++i) {
    if (vocab_offsets[i + 1] < vocab_offsets[i]) {
      delete t;
      return amdr::fail(AMDR_EINVAL, "tokenizer_create: offsets not ascending at term %lld", (long long)i);
    }
  }
  if (n_terms) {
    const int64_t base = vocab_offsets[0];
    t->blob.assign(vocab_blob + base, (size_t)(vocab_offsets[n_terms] - base));
    t->offs.resize((size_t)n_terms + 1);
    for (int64_t i = 0; i <= n_terms; ++i) t->offs[(size_t)i] = vocab_offsets[i] - base;
    size_t cap = 16;
    while (cap < (size_t)n_terms * 2) cap <<= 1;
    t->slots.assign(cap, -1);
    t->mask = (uint32_t)(cap - 1);
    for (int64_t i = 0; i < n_terms; ++i) {  // first id of a repeated term wins
      const unsigned char* p = reinterpret_cast<const unsigned char*>(t->blob.data()) + t->offs[(size_t)i];
      const size_t n = (size_t)(t->offs[(size_t)i + 1] - t->offs[(size_t)i]);
      if (t->find_slow(p, n) >= 0) continue;
      uint32_t j = amdr_tokenizer::hash(p, n) & t->mask;
      while (t->slots[j] >= 0) j = (j + 1) & t->mask;
      t->slots[j] = (int32_t)i;
      if (n == 1) t->single[p[0]] = (int32_t)i;
    }
  }
  *out = t;
  return AMDR_OK;
}

}  // extern "C"

namespace {

// A small persistent pool: a batch of tens of thousands of queries is cut into ranges, one per worker; starting
// std::threads per call cost more than tokenising a few thousand queries.  Workers sleep on a condition variable
// between calls; calls from several Python threads are serialised on the pool (each still runs on all workers).
class Pool {
 public:
  static Pool& get() {
    // never destroyed: no join at process exit (detached daemon threads).  A forked child has none of the parent's
    // threads: it starts its own pool at its first batch.
    static std::once_flag once;
    std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { inst().store(nullptr); }); });
    Pool* p = inst().load();
    if (!p) {
      static std::mutex mk;
      std::lock_guard<std::mutex> g(mk);
      p = inst().load();
      if (!p) {
        p = new Pool();
        inst().store(p);
      }
    }
    return *p;
  }
  static std::atomic<Pool*>& inst() {
    static std::atomic<Pool*> i{nullptr};
    return i;
  }
  int workers() const { return (int)th_.size() + 1; }
  // fn(part) for part = 0 .. parts - 1, parts <= workers(); the caller runs part 0
  void run(int parts, const std::function<void(int)>& fn) {
    if (parts <= 1) {
      fn(0);
      return;
    }
    std::lock_guard<std::mutex> call(call_mu_);
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = &fn;
      parts_ = parts;
      pending_ = parts - 1;
      ++gen_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> g(mu_);
    done_.wait(g, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  Pool() {
    int n = (int)std::thread::hardware_concurrency();
    const char* e = getenv("AMDR_TOKENIZER_THREADS");
    if (e && atoi(e) >= 1) n = atoi(e);
    if (n > 32) n = 32;
    if (n < 1) n = 1;
    for (int i = 1; i < n; ++i) {
      th_.emplace_back([this, i] { loop(i); });
      th_.back().detach();
    }
  }
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (id < parts_) fn = fn_;
      }
      if (fn) {
        (*fn)(id);
        std::lock_guard<std::mutex> g(mu_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int parts_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
};

// the batch core: query q = the lens[q] bytes at ptrs[q] -> CSR.  Two phases: every worker tokenises its range of
// queries into its own term buffer (+ per-query counts), then the prefix sum over the counts gives q_ptr and every
// worker copies its terms to their place.
int encode_core(const amdr_tokenizer* t, const unsigned char* const* ptrs, const int64_t* lens, int32_t nq,
                int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter) {
  if (nq == 0) return AMDR_OK;
  q_ptr[0] = 0;
  Pool& pool = Pool::get();
  int parts = pool.workers();
  if (parts > nq / 256) parts = nq / 256;  // a worker is worth waking for a few hundred queries
  if (parts < 1) parts = 1;
  std::vector<std::vector<int32_t>> bufs((size_t)parts);
  std::atomic<int> bad{-1};
  auto range = [&](int p, int32_t* lo, int32_t* hi) {
    *lo = (int32_t)((int64_t)nq * p / parts);
    *hi = (int32_t)((int64_t)nq * (p + 1) / parts);
  };
  std::function<void(int)> phase1 = [&](int p) {
    int32_t lo, hi;
    range(p, &lo, &hi);
    std::vector<int32_t>& out = bufs[(size_t)p];
    int64_t bytes = 0;
    for (int32_t q = lo; q < hi; ++q) bytes += lens[q] > 0 ? lens[q] : 0;
    out.reserve((size_t)(bytes / 2 + 16));
    for (int32_t q = lo; q < hi; ++q) {
      const int64_t n = lens[q];
      if (n < 0 || n >= (1ll << 31) || (n > 0 && !ptrs[q])) {
        bad.store(q);
        return;
      }
      const unsigned char* s = ptrs[q];
      const size_t start = out.size();
      const bool ok = tokenize(s, (int)n, [&](int x, int y) { out.push_back(t->find(s + x, (size_t)(y - x))); });
      if (!ok) out.resize(start);
      needs_segmenter[q] = ok ? 0 : 1;
      q_ptr[q + 1] = (int64_t)(out.size() - start);  // the count; turned into the offset below
    }
  };
  pool.run(parts, phase1);
  AMDR_REQUIRE(bad.load() < 0, "tokenizer_encode: bad text at query %d", bad.load());
  for (int32_t q = 0; q < nq; ++q) q_ptr[q + 1] += q_ptr[q];
  AMDR_REQUIRE(q_ptr[nq] <= capacity, "tokenizer_encode: term buffer too small (capacity %lld, %lld terms)",
               (long long)capacity, (long long)q_ptr[nq]);
  std::function<void(int)> phase2 = [&](int p) {
    int32_t lo, hi;
    range(p, &lo, &hi);
    const std::vector<int32_t>& src = bufs[(size_t)p];
    if (!src.empty()) memcpy(term_ids + q_ptr[lo], src.data(), src.size() * sizeof(int32_t));
  };
  pool.run(parts, phase2);
  return AMDR_OK;
}

// offsets into one blob -> pointer / length arrays (`trim` bytes of separator behind every query but the last)
int encode_offsets(const amdr_tokenizer* t, const unsigned char* text, const int64_t* offs, int64_t trim, int32_t nq,
                   int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter) {
  std::vector<const unsigned char*> ptrs((size_t)nq);
  std::vector<int64_t> lens((size_t)nq);
  for (int32_t q = 0; q < nq; ++q) {
    ptrs[(size_t)q] = text + offs[q];
    lens[(size_t)q] = offs[q + 1] - offs[q] - (q + 1 < nq ? trim : 0);
  }
  return encode_core(t, ptrs.data(), lens.data(), nq, term_ids, capacity, q_ptr, needs_segmenter);
}

}  // namespace

extern "C" {

int amdr_tokenizer_encode(const amdr_tokenizer_t* t, const char* text_blob, const int64_t* text_offsets, int32_t nq,
                          int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter) {
  AMDR_REQUIRE(t != nullptr, "tokenizer_encode: null handle");
  AMDR_REQUIRE(nq >= 0 && (nq == 0 || (text_offsets && q_ptr && needs_segmenter)), "tokenizer_encode: null buffer");
  AMDR_REQUIRE(capacity >= 0 && (capacity == 0 || term_ids), "tokenizer_encode: null term buffer");
  return encode_offsets(t, reinterpret_cast<const unsigned char*>(text_blob), text_offsets, 0, nq, term_ids, capacity, q_ptr,
                        needs_segmenter);
}

int amdr_tokenizer_encode_ptrs(const amdr_tokenizer_t* t, const char* const* texts, const int64_t* n_bytes, int32_t nq,
                               int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter) {
  AMDR_REQUIRE(t != nullptr, "tokenizer_encode_ptrs: null handle");
  AMDR_REQUIRE(nq >= 0 && (nq == 0 || (texts && n_bytes && q_ptr && needs_segmenter)), "tokenizer_encode_ptrs: null buffer");
  AMDR_REQUIRE(capacity >= 0 && (capacity == 0 || term_ids), "tokenizer_encode_ptrs: null term buffer");
  return encode_core(t, reinterpret_cast<const unsigned char* const*>(texts), n_bytes, nq, term_ids, capacity, q_ptr,
                     needs_segmenter);
}

int amdr_tokenizer_encode_joined(const amdr_tokenizer_t* t, const char* text_blob, int64_t n_bytes, int32_t nq,
                                 int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter) {
  AMDR_REQUIRE(t != nullptr, "tokenizer_encode_joined: null handle");
  AMDR_REQUIRE(nq >= 0 && n_bytes >= 0 && (nq == 0 || (q_ptr && needs_segmenter)), "tokenizer_encode_joined: null buffer");
  AMDR_REQUIRE(n_bytes == 0 || text_blob, "tokenizer_encode_joined: null text");
  AMDR_REQUIRE(capacity >= 0 && (capacity == 0 || term_ids), "tokenizer_encode_joined: null term buffer");
  if (nq == 0) return AMDR_OK;
  // queries are separated by ONE NUL byte (nq - 1 of them): offsets from a memchr walk
  std::vector<int64_t> offs((size_t)nq + 1);
  offs[0] = 0;
  const char* p = text_blob;
  const char* end = text_blob + n_bytes;
  for (int32_t q = 1; q < nq; ++q) {
    const char* z = p < end ? static_cast<const char*>(memchr(p, 0, (size_t)(end - p))) : nullptr;
    AMDR_REQUIRE(z != nullptr, "tokenizer_encode_joined: %d queries announced, separator %d missing", nq, q);
    offs[(size_t)q] = (z - text_blob) + 1;
    p = z + 1;
  }
  AMDR_REQUIRE(p > end || memchr(p, 0, (size_t)(end - p)) == nullptr, "tokenizer_encode_joined: more separators than queries");
  offs[(size_t)nq] = n_bytes;
  return encode_offsets(t, reinterpret_cast<const unsigned char*>(text_blob), offs.data(), 1, nq, term_ids, capacity, q_ptr,
                        needs_segmenter);
}

int amdr_tokenizer_spans(const char* text, int64_t n_bytes, int32_t* starts, int32_t* ends, int32_t capacity,
                         int32_t* n_tokens) {
  AMDR_REQUIRE(n_tokens != nullptr && n_bytes >= 0 && n_bytes < (1ll << 31), "tokenizer_spans: bad arguments");
  AMDR_REQUIRE(n_bytes == 0 || text, "tokenizer_spans: null text");
  AMDR_REQUIRE(capacity >= 0 && (capacity == 0 || (starts && ends)), "tokenizer_spans: null span buffers");
  int32_t n = 0;
  bool overflow = false;
  const bool ok = tokenize(reinterpret_cast<const unsigned char*>(text), (int)n_bytes, [&](int a, int b) {
    if (n >= capacity) {
      overflow = true;
      return;
    }
    starts[n] = a;
    ends[n] = b;
    ++n;
  });
  AMDR_REQUIRE(!overflow, "tokenizer_spans: span buffers too small");
  *n_tokens = ok ? n : -1;  // -1: the text holds a Han character and needs a segmenter
  return AMDR_OK;
}

int amdr_tokenizer_destroy(amdr_tokenizer_t* t) {
  delete t;
  return AMDR_OK;
}

}  // extern "C"
