// Host-side caption tokeniser (no device code): byte-level BPE of the CLIP family, batch-parallel on std::thread.
// Replaces the per-caption Python loop of model/utils/text_utils/simple_tokenizer.py:82-129 (regex pre-tokenisation,
// `bpe()` merge loop, id lookup) and the padding / truncation of text_encoder/text_transformer.py:144-180, which at
// > 10^4 captions/s per GPU is the first thing on the host to fall behind.  Written from the published
// algorithm; results are pinned bit-exactly to the Python tokeniser (declip_amd/bpe.py, itself pinned to the reference's) by
// tests/test_tokenizer.py.
//
// Scope: captions that are pure ASCII after cleaning / lower-casing (the caller cleans: ftfy, html.unescape, whitespace
// collapse, lower -- cheap C-implemented Python builtins).  The pre-tokenisation regex uses Unicode classes (\p{L}, \p{N});
// for ASCII they are [a-z] / [0-9].  A caption with any other byte is NOT tokenised here: its status is set to 1 and the
// caller runs the Python path for that row, so parity never depends on a re-implemented Unicode table.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dh_common.h"

namespace {

struct PairHash {
  size_t operator()(const uint64_t& k) const { return (size_t)(k * 0x9E3779B97F4A7C15ull >> 17); }
};

struct Bpe {
  int byte_id[256];                                            // byte -> id of its printable stand-in (bytes_to_unicode order)
  std::unordered_map<uint64_t, int32_t, PairHash> rank;        // (left id << 32 | right id) -> merge rank
  std::vector<int32_t> merged_id;                              // rank -> id of the merged token
  int32_t sot = 0, eot = 0, mask = 0, vocab = 0;
  static constexpr int NSHARD = 64;
  std::mutex mu[NSHARD];
  std::unordered_map<std::string, std::vector<int32_t>> cache[NSHARD];
};

// utf-8 decode of one code point (merges file: stand-in characters are U+0021..U+0143)
inline bool next_cp(const uint8_t*& p, const uint8_t* e, uint32_t* cp) {
  if (p >= e) return false;
  uint8_t c = *p++;
  if (c < 0x80) { *cp = c; return true; }
  int n = (c >= 0xF0) ? 3 : (c >= 0xE0) ? 2 : 1;
  uint32_t v = c & (0x3F >> n);
  while (n-- && p < e) v = (v << 6) | (*p++ & 0x3F);
  *cp = v;
  return true;
}

inline bool is_letter(uint8_t c) { return c >= 'a' && c <= 'z'; }     // input is lower-cased
inline bool is_upper(uint8_t c) { return c >= 'A' && c <= 'Z'; }
inline bool is_digit(uint8_t c) { return c >= '0' && c <= '9'; }
inline bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// the BPE merge loop on ids: repeatedly merge every occurrence of the lowest-ranked adjacent pair
void bpe_word(Bpe& t, const uint8_t* w, int n, std::vector<int32_t>& out) {
  std::string key(reinterpret_cast<const char*>(w), n);
  const int sh = (int)(std::hash<std::string>()(key) % Bpe::NSHARD);
  {
    std::lock_guard<std::mutex> g(t.mu[sh]);
    auto it = t.cache[sh].find(key);
    if (it != t.cache[sh].end()) { out.insert(out.end(), it->second.begin(), it->second.end()); return; }
  }
  std::vector<int32_t> sym(n);
  for (int i = 0; i < n; ++i) sym[i] = t.byte_id[w[i]];
  sym[n - 1] += 256;                                           // last symbol carries the end-of-word marker "</w>"
  while (sym.size() > 1) {
    int32_t best = INT32_MAX;
    uint64_t best_key = 0;
    for (size_t i = 0; i + 1 < sym.size(); ++i) {
      const uint64_t k = ((uint64_t)(uint32_t)sym[i] << 32) | (uint32_t)sym[i + 1];
      auto it = t.rank.find(k);
      if (it != t.rank.end() && it->second < best) { best = it->second; best_key = k; }
    }
    if (best == INT32_MAX) break;
    const int32_t a = (int32_t)(best_key >> 32), b = (int32_t)(best_key & 0xffffffffu), m = t.merged_id[best];
    size_t o = 0;
    for (size_t i = 0; i < sym.size();) {
      if (i + 1 < sym.size() && sym[i] == a && sym[i + 1] == b) { sym[o++] = m; i += 2; }
      else sym[o++] = sym[i++];
    }
    sym.resize(o);
  }
  out.insert(out.end(), sym.begin(), sym.end());
  std::lock_guard<std::mutex> g(t.mu[sh]);
  if (t.cache[sh].size() < (1u << 20)) t.cache[sh].emplace(std::move(key), std::move(sym));
}

// one caption -> ids (no SOT / EOT).  Returns false when the caption needs the Unicode-aware path.
bool encode_ascii(Bpe& t, const uint8_t* s, int n, std::vector<int32_t>& ids) {
  for (int i = 0; i < n; ++i) {
    const uint8_t c = s[i];
    if (c >= 0x7f || (c < 0x20 && !is_space(c)) || is_upper(c)) return false;
  }
  static const char* CONTR[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};      // alternation order of the pattern
  int i = 0;
  while (i < n) {
    const uint8_t c = s[i];
    if (is_space(c)) { ++i; continue; }
    if (c == '<') {
      if (n - i >= 15 && memcmp(s + i, "<|startoftext|>", 15) == 0) { ids.push_back(t.sot); i += 15; continue; }
      if (n - i >= 13 && memcmp(s + i, "<|endoftext|>", 13) == 0) { ids.push_back(t.eot); i += 13; continue; }
    }
    int len = 0;
    if (c == '\'') {
      for (const char* k : CONTR) {
        const int kl = (int)strlen(k);
        if (n - i >= kl && memcmp(s + i, k, kl) == 0) { len = kl; break; }
      }
    }
    if (!len) {
      if (is_letter(c)) { while (i + len < n && is_letter(s[i + len])) ++len; }
      else if (is_digit(c)) len = 1;
      else { while (i + len < n && !is_space(s[i + len]) && !is_letter(s[i + len]) && !is_digit(s[i + len])) ++len; }
    }
    bpe_word(t, s + i, len, ids);
    i += len;
  }
  return true;
}

}  // namespace

// merges: the decompressed text of bpe_simple_vocab_16e6.txt(.gz) -- a header line, then one "left right" merge per line;
// the first n_merges of them are used (simple_tokenizer.py:66-69: 49152 - 256 - 2).  Vocabulary order as in the reference:
// 256 byte stand-ins, the same with "</w>", the merges, then <|mask|>, <|startoftext|>, <|endoftext|>.
extern "C" void* dh_bpe_create(const char* merges, int64_t nbytes, int n_merges) {
  if (!merges || nbytes <= 0 || n_merges < 0) { dh_set_error("dh_bpe_create: bad args"); return nullptr; }
  Bpe* t = new Bpe();
  // bytes_to_unicode(): printable bytes keep their code point, the others get 256, 257, ... in increasing byte order
  std::vector<int> keep;
  for (int b = 33; b <= 126; ++b) keep.push_back(b);
  for (int b = 161; b <= 172; ++b) keep.push_back(b);
  for (int b = 174; b <= 255; ++b) keep.push_back(b);
  std::vector<uint32_t> cp_of(256);
  {
    std::vector<bool> kept(256, false);
    for (int b : keep) kept[b] = true;
    int idx = 0, extra = 0;
    for (int b : keep) { t->byte_id[b] = idx++; cp_of[b] = b; }
    for (int b = 0; b < 256; ++b)
      if (!kept[b]) { t->byte_id[b] = idx++; cp_of[b] = 256 + extra++; }
  }
  // token string -> id, later entries win (the reference builds a dict over the vocabulary list)
  std::unordered_map<std::string, int32_t> enc;
  auto utf8 = [](uint32_t cp, std::string& s) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
  };
  for (int b = 0; b < 256; ++b) {
    std::string s;
    utf8(cp_of[b], s);
    enc[s] = t->byte_id[b];
  }
  for (int b = 0; b < 256; ++b) {
    std::string s;
    utf8(cp_of[b], s);
    enc[s + "</w>"] = 256 + t->byte_id[b];
  }
  const char* p = merges;
  const char* e = merges + nbytes;
  while (p < e && *p != '\n') ++p;                             // header line
  if (p < e) ++p;
  struct M { std::string a, b; };
  std::vector<M> ms;
  while (p < e && (int)ms.size() < n_merges) {
    const char* q = p;
    while (q < e && *q != '\n') ++q;
    // split on whitespace (str.split())
    std::vector<std::string> parts;
    const char* a = p;
    while (a < q) {
      while (a < q && (*a == ' ' || *a == '\t' || *a == '\r')) ++a;
      const char* b2 = a;
      while (b2 < q && *b2 != ' ' && *b2 != '\t' && *b2 != '\r') ++b2;
      if (b2 > a) parts.emplace_back(a, b2 - a);
      a = b2;
    }
    M m;
    if (parts.size() >= 1) m.a = parts[0];
    if (parts.size() >= 2) m.b = parts[1];
    if (parts.size() != 2) { m.a.clear(); m.b.clear(); }       // a malformed line still occupies a rank / an id
    ms.push_back(m);
    p = q < e ? q + 1 : q;
  }
  if ((int)ms.size() != n_merges) { dh_set_error("dh_bpe_create: merges file has %d usable lines, %d requested", (int)ms.size(), n_merges); delete t; return nullptr; }
  for (int r = 0; r < n_merges; ++r) enc[ms[r].a + ms[r].b] = 512 + r;
  t->mask = 512 + n_merges; t->sot = t->mask + 1; t->eot = t->mask + 2; t->vocab = t->mask + 3;
  t->merged_id.resize(n_merges);
  for (int r = 0; r < n_merges; ++r) {
    t->merged_id[r] = enc[ms[r].a + ms[r].b];
    auto ia = enc.find(ms[r].a), ib = enc.find(ms[r].b);
    if (ms[r].a.empty() || ia == enc.end() || ib == enc.end()) continue;                 // can never apply
    t->rank[((uint64_t)(uint32_t)ia->second << 32) | (uint32_t)ib->second] = r;          // later duplicates win, as in a dict
  }
  return t;
}

extern "C" void dh_bpe_destroy(void* h) { delete static_cast<Bpe*>(h); }

extern "C" int dh_bpe_vocab_size(void* h) { return h ? static_cast<Bpe*>(h)->vocab : 0; }

// texts: n cleaned, lower-cased captions concatenated; caption i = bytes [offsets[i], offsets[i+1]).
// out[n][ctx] int64: SOT, ids, EOT, zero padding; over-long captions keep the first ctx-1 tokens and the final EOT
// (text_transformer.py:150-155).  status[i]: 0 = written, 1 = not ASCII -> the caller tokenises that caption itself.
extern "C" int dh_bpe_encode(void* h, const char* texts, const int64_t* offsets, int n, int ctx, int64_t* out, int32_t* status,
                             int n_threads) {
  DH_REQUIRE(h && texts && offsets && out && status && n >= 0 && ctx >= 2, "dh_bpe_encode: bad args");
  Bpe& t = *static_cast<Bpe*>(h);
  auto work = [&](int lo, int hi) {
    std::vector<int32_t> ids;
    for (int i = lo; i < hi; ++i) {
      ids.clear();
      ids.push_back(t.sot);
      const bool ok = encode_ascii(t, reinterpret_cast<const uint8_t*>(texts) + offsets[i], (int)(offsets[i + 1] - offsets[i]), ids);
      status[i] = ok ? 0 : 1;
      int64_t* row = out + (int64_t)i * ctx;
      for (int k = 0; k < ctx; ++k) row[k] = 0;
      if (!ok) continue;
      ids.push_back(t.eot);
      const int m = (int)ids.size();
      if (m > ctx) {
        for (int k = 0; k < ctx - 1; ++k) row[k] = ids[k];
        row[ctx - 1] = ids[m - 1];
      } else {
        for (int k = 0; k < m; ++k) row[k] = ids[k];
      }
    }
  };
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  if (n_threads == 1 || n < 2 * n_threads) { work(0, n); return DH_OK; }
  std::vector<std::thread> th;
  const int per = (n + n_threads - 1) / n_threads;
  for (int k = 0; k < n_threads; ++k) {
    const int lo = k * per, hi = std::min(n, lo + per);
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& x : th) x.join();
  return DH_OK;
}
