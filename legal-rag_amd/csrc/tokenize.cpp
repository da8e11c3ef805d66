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
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"

struct amdr_tokenizer {
  std::unordered_map<std::string, int32_t> vocab;
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

// tokens of one sentence; returns false (nothing emitted is meaningful) when it holds a Han character
template <class Emit>
inline bool tokenize(const unsigned char* s, int n, Emit&& emit) {
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
  AMDR_REQUIRE(n_terms >= 0 && n_terms < (1ll << 31) && (n_terms == 0 || (vocab_blob && vocab_offsets)),
               "tokenizer_create: bad vocabulary");
  amdr_tokenizer* t = new (std::nothrow) amdr_tokenizer();
  if (!t) return amdr::fail(AMDR_ENOMEM, "tokenizer_create: host alloc");
  t->vocab.reserve((size_t)n_terms * 2);
  for (int64_t i = 0; i < n_terms; ++i) {
    const int64_t lo = vocab_offsets[i], hi = vocab_offsets[i + 1];
    if (hi < lo) {
      delete t;
      return amdr::fail(AMDR_EINVAL, "tokenizer_create: offsets not ascending at term %lld", (long long)i);
    }
    t->vocab.emplace(std::string(vocab_blob + lo, (size_t)(hi - lo)), (int32_t)i);  // first id of a repeated term wins
  }
  *out = t;
  return AMDR_OK;
}

int amdr_tokenizer_encode(const amdr_tokenizer_t* t, const char* text_blob, const int64_t* text_offsets, int32_t nq,
                          int32_t* term_ids, int64_t capacity, int64_t* q_ptr, int32_t* needs_segmenter) {
  AMDR_REQUIRE(t != nullptr, "tokenizer_encode: null handle");
  AMDR_REQUIRE(nq >= 0 && (nq == 0 || (text_offsets && q_ptr && needs_segmenter)), "tokenizer_encode: null buffer");
  AMDR_REQUIRE(capacity >= 0 && (capacity == 0 || term_ids), "tokenizer_encode: null term buffer");
  int64_t at = 0;
  std::string key;
  if (nq) q_ptr[0] = 0;
  for (int32_t q = 0; q < nq; ++q) {
    const int64_t lo = text_offsets[q], hi = text_offsets[q + 1];
    AMDR_REQUIRE(hi >= lo && hi - lo < (1ll << 31), "tokenizer_encode: bad offsets at query %d", q);
    const unsigned char* s = reinterpret_cast<const unsigned char*>(text_blob) + lo;
    const int64_t start = at;
    bool overflow = false;
    const bool ok = tokenize(s, (int)(hi - lo), [&](int a, int b) {
      if (at >= capacity) {
        overflow = true;
        return;
      }
      key.assign(reinterpret_cast<const char*>(s) + a, (size_t)(b - a));
      auto it = t->vocab.find(key);
      term_ids[at++] = it == t->vocab.end() ? -1 : it->second;
    });
    AMDR_REQUIRE(!overflow, "tokenizer_encode: term buffer too small (capacity %lld)", (long long)capacity);
    if (!ok) at = start;
    needs_segmenter[q] = ok ? 0 : 1;
    q_ptr[q + 1] = at;
  }
  return AMDR_OK;
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
