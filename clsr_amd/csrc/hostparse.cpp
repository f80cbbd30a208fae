// Host-side tokenizer of the sequential TSV format (reference io/sequential_iterator.py:90-163, parser_one_line):
//   label \t user \t item \t cate \t ts \t item_hist(csv) \t cate_hist(csv) \t ts_hist(csv)
// One pass over the file bytes: vocabulary look-ups through an open-addressing hash of the pickled dict's keys
// (FNV-1a over the UTF-8 bytes), numbers through strtol / strtod, histories flattened with per-line offsets.
// The Python side (clsr_amd/sequential_iterator.py) computes the time features and the padding from these arrays
// with the same numpy expressions as the literal parser; anything irregular (short lines, ragged history columns,
// number syntax strtod does not fully consume) returns non-zero and the caller falls back to the literal parser.
// Plain C++ (no device code): part of libclsr_hip.so so that one library carries the whole C ABI.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "clsr_hip.h"

namespace {
struct Vocab {
  std::vector<char> blob;
  std::vector<long> off;     // n + 1
  std::vector<int> ids;
  std::vector<long> slot;    // open addressing: token index or -1
  unsigned long mask;
};

inline unsigned long fnv1a(const char* p, long n) {
  unsigned long h = 1469598103934665603ul;
  for (long i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ul; }
  return h;
}

inline int lookup(const Vocab* v, const char* p, long n) {
  unsigned long s = fnv1a(p, n) & v->mask;
  for (;;) {
    const long t = v->slot[s];
    if (t < 0) return 0;                                   // dict.get(token, 0)
    const long len = v->off[t + 1] - v->off[t];
    if (len == n && memcmp(v->blob.data() + v->off[t], p, (size_t)n) == 0) return v->ids[t];
    s = (s + 1) & v->mask;
  }
}

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// [b, e) with leading / trailing whitespace removed (str.strip())
inline void strip(const char*& b, const char*& e) {
  while (b < e && is_space(*b)) ++b;
  while (e > b && is_space(e[-1])) --e;
}

inline bool parse_double(const char* b, const char* e, double* out) {
  if (b >= e || e - b > 63) return false;
  char tmp[64];
  memcpy(tmp, b, (size_t)(e - b));
  tmp[e - b] = 0;
  for (const char* p = tmp; *p; ++p)                       // plain decimal / exponent syntax only
    if (!((*p >= '0' && *p <= '9') || *p == '.' || *p == '-' || *p == '+' || *p == 'e' || *p == 'E')) return false;
  char* end = nullptr;
  *out = strtod(tmp, &end);
  return end == tmp + (e - b);
}

inline bool parse_int(const char* b, const char* e, int* out) {
  if (b >= e || e - b > 20) return false;
  char tmp[24];
  memcpy(tmp, b, (size_t)(e - b));
  tmp[e - b] = 0;
  for (const char* p = tmp; *p; ++p)
    if (!((*p >= '0' && *p <= '9') || *p == '-' || *p == '+')) return false;
  char* end = nullptr;
  const long v = strtol(tmp, &end, 10);
  if (end != tmp + (e - b)) return false;
  *out = (int)v;
  return true;
}
}  // namespace

extern "C" void* clsr_host_vocab_create(const char* blob, const long* offsets, const int* ids, long n) {
  if (!blob || !offsets || !ids || n < 0) return nullptr;
  Vocab* v = new Vocab();
  v->blob.assign(blob, blob + offsets[n]);
  v->off.assign(offsets, offsets + n + 1);
  v->ids.assign(ids, ids + n);
  unsigned long cap = 16;
  while (cap < (unsigned long)(2 * n + 1)) cap <<= 1;
  v->mask = cap - 1;
  v->slot.assign(cap, -1);
  for (long t = 0; t < n; ++t) {
    unsigned long s = fnv1a(v->blob.data() + v->off[t], v->off[t + 1] - v->off[t]) & v->mask;
    while (v->slot[s] >= 0) s = (s + 1) & v->mask;
    v->slot[s] = t;
  }
  return v;
}

extern "C" int clsr_host_vocab_destroy(void* vocab) {
  delete static_cast<Vocab*>(vocab);
  return 0;
}

// pass 1: number of lines and of item-history tokens (commas + 1 in the 6th field)
extern "C" int clsr_host_tsv_count(const char* buf, long nbytes, long* n_lines, long* n_tokens) {
  if (!buf || !n_lines || !n_tokens) return -1;
  long lines = 0, toks = 0;
  const char* p = buf;
  const char* end = buf + nbytes;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    const char* lb = p;
    strip(lb, le);                                         // line.strip() before the split
    int field = 0;
    long t = 1;
    for (const char* c = lb; c < le; ++c) {
      if (*c == '\t') { if (++field > 5) break; }
      else if (field == 5 && *c == ',') ++t;
    }
    if (field < 5) return 1;                               // short line: the literal parser decides
    toks += t;
    ++lines;
    p = nl ? nl + 1 : end;
  }
  *n_lines = lines;
  *n_tokens = toks;
  return 0;
}

// pass 2: fill the arrays; hist_off has n_lines + 1 entries.  Returns 0, or 1 when a line is irregular.
extern "C" int clsr_host_tsv_parse(const char* buf, long nbytes, const void* user_vocab, const void* item_vocab,
                                   const void* cate_vocab, int* labels, int* users, int* items, int* cates,
                                   double* cur_time, long* hist_off, int* hist_items, int* hist_cates,
                                   double* hist_ts) {
  if (!buf || !user_vocab || !item_vocab || !cate_vocab) return -1;
  const Vocab* uv = static_cast<const Vocab*>(user_vocab);
  const Vocab* iv = static_cast<const Vocab*>(item_vocab);
  const Vocab* cv = static_cast<const Vocab*>(cate_vocab);
  const char* p = buf;
  const char* end = buf + nbytes;
  long line = 0, tok = 0;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* lb = p;
    const char* le = nl ? nl : end;
    p = nl ? nl + 1 : end;
    strip(lb, le);                                         // line.strip()
    const char* fb[8];
    const char* fe[8];
    int nf = 0;
    const char* s = lb;
    for (const char* c = lb; c <= le && nf < 8; ++c)
      if (c == le || *c == '\t') {
        fb[nf] = s; fe[nf] = c; ++nf; s = c + 1;
        if (c == le) break;
      }
    if (nf < 8) return 1;
    if (!parse_int(fb[0], fe[0], &labels[line])) return 1;
    users[line] = lookup(uv, fb[1], fe[1] - fb[1]);
    items[line] = lookup(iv, fb[2], fe[2] - fb[2]);
    cates[line] = lookup(cv, fb[3], fe[3] - fb[3]);
    if (!parse_double(fb[4], fe[4], &cur_time[line])) return 1;
    hist_off[line] = tok;
    long cnt[3] = {0, 0, 0};
    for (int col = 0; col < 3; ++col) {
      const char* hb = fb[5 + col];
      const char* he = fe[5 + col];
      strip(hb, he);                                       // words[k].strip()
      const char* ts_ = hb;
      long k = 0;
      for (const char* c = hb; c <= he; ++c)
        if (c == he || *c == ',') {
          const long at = tok + k;
          if (col == 0) hist_items[at] = lookup(iv, ts_, c - ts_);
          else if (col == 1) { if (k >= cnt[0]) return 1; hist_cates[at] = lookup(cv, ts_, c - ts_); }
          else { if (k >= cnt[0]) return 1; if (!parse_double(ts_, c, &hist_ts[at])) return 1; }
          ++k;
          ts_ = c + 1;
          if (c == he) break;
        }
      cnt[col] = k;
    }
    if (cnt[1] != cnt[0] || cnt[2] != cnt[0]) return 1;    // ragged history columns
    tok += cnt[0];
    ++line;
  }
  hist_off[line] = tok;
  return 0;
}
