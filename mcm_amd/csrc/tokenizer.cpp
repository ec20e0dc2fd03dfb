// tokenizer.cpp — CLIP byte-level BPE tokenizer, host side.
//
// Replaces the reference's `CLIPTokenizer.from_pretrained(args.ckpt)` +
// `tokenizer(list[str], padding=True, return_tensors="pt")` (reference
// utils/detection_util.py:216,228): prompts -> [n, S] token ids + attention mask, BOS/EOS added,
// right-padded with the pad token to the longest prompt.  The algorithm is the one HF transformers'
// CLIPTokenizer configures (third-party, transformers 5.15 here; tokenization_clip.py __init__):
//   normalise   NFC (input is assumed precomposed), whitespace runs -> one space, lowercase
//   split       <|startoftext|> | <|endoftext|> | 's|'t|'re|'ve|'m|'ll|'d | \p{L}+ | \p{N} | [^\s\p{L}\p{N}]+
//   byte level  UTF-8 bytes -> the GPT-2 printable-character alphabet
//   BPE         symbols of a word, "</w>" appended to the last one, merged by rank; unknown -> unk
// The vocabulary (vocab.json, merges.txt of the checkpoint) is read from files given by the caller;
// neither file exists in the build containers, so parity is pinned on synthetic vocabularies against
// HF's tokenizer built from the same vocabulary (tests/test_tokenizer_bpe.py).
// Unicode scope: letter / number / space classes and lowercasing cover ASCII, Latin-1, Latin
// Extended-A/B, IPA, Greek, Cyrillic, Hebrew, Arabic, Devanagari, CJK, kana and Hangul blocks (class
// names and prompt templates); other code points are treated as symbols.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mcm.h"

struct mcm_tokenizer {
  std::unordered_map<std::string, int32_t> vocab;
  std::unordered_map<std::string, int32_t> rank;  // "left right" -> merge rank
  std::unordered_map<std::string, std::vector<int32_t>> cache;
  std::string b2u[256];                           // byte -> UTF-8 of its printable stand-in
  int32_t bos = -1, eos = -1, unk = -1, pad = -1;
  std::string err;
};

namespace {

void put_utf8(std::string& s, uint32_t c) {
  if (c < 0x80) s += (char)c;
  else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
  else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
  else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
}

// next code point of a UTF-8 string (invalid bytes pass through as Latin-1)
uint32_t next_cp(const std::string& s, size_t& i) {
  const unsigned char c = (unsigned char)s[i];
  auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
  if (c < 0x80) { i += 1; return c; }
  if ((c & 0xE0) == 0xC0 && cont(1)) { const uint32_t v = ((c & 0x1F) << 6) | (s[i + 1] & 0x3F); i += 2; return v; }
  if ((c & 0xF0) == 0xE0 && cont(1) && cont(2)) {
    const uint32_t v = ((c & 0x0F) << 12) | ((s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F); i += 3; return v;
  }
  if ((c & 0xF8) == 0xF0 && cont(1) && cont(2) && cont(3)) {
    const uint32_t v = ((c & 0x07) << 18) | ((s[i + 1] & 0x3F) << 12) | ((s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F);
    i += 4; return v;
  }
  i += 1;
  return c;
}

bool is_space(uint32_t c) {  // \s of the regex engine
  return c == ' ' || (c >= 0x09 && c <= 0x0D) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
         (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
bool is_number(uint32_t c) {  // \p{N}
  return (c >= '0' && c <= '9') || c == 0xB2 || c == 0xB3 || c == 0xB9 || (c >= 0xBC && c <= 0xBE) ||
         (c >= 0x0660 && c <= 0x0669) || (c >= 0x06F0 && c <= 0x06F9) || (c >= 0x0966 && c <= 0x096F) ||
         (c >= 0x2070 && c <= 0x2079) || (c >= 0x2080 && c <= 0x2089) || (c >= 0x2150 && c <= 0x218B) ||
         (c >= 0x2460 && c <= 0x249B) || (c >= 0xFF10 && c <= 0xFF19);
}
bool is_letter(uint32_t c) {  // \p{L}
  if (c < 0x80) return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
  if (c < 0x100) return c == 0xAA || c == 0xB5 || c == 0xBA || (c >= 0xC0 && c != 0xD7 && c != 0xF7);
  return (c >= 0x0100 && c <= 0x02C1) || (c >= 0x0370 && c <= 0x03FF && c != 0x037E && c != 0x0387 && c != 0x03F6) ||
         (c >= 0x0400 && c <= 0x0481) || (c >= 0x048A && c <= 0x052F) || (c >= 0x0531 && c <= 0x0556) ||
         (c >= 0x0561 && c <= 0x0587) || (c >= 0x05D0 && c <= 0x05EA) || (c >= 0x0620 && c <= 0x064A) ||
         (c >= 0x0671 && c <= 0x06D3) || (c >= 0x0904 && c <= 0x0939) || (c >= 0x1E00 && c <= 0x1FFC) ||
         (c >= 0x3041 && c <= 0x3096) || (c >= 0x30A1 && c <= 0x30FA) || (c >= 0x3400 && c <= 0x4DBF) ||
         (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0xAC00 && c <= 0xD7A3) || (c >= 0xFF21 && c <= 0xFF3A) ||
         (c >= 0xFF41 && c <= 0xFF5A);
}
uint32_t lower(uint32_t c) {
  if (c >= 'A' && c <= 'Z') return c + 32;
  if (c < 0xC0) return c;
  if (c <= 0xDE && c != 0xD7) return c + 32;
  if (c >= 0x0100 && c <= 0x017F) {  // Latin Extended-A: pairs, with the two shifted runs
    if (c == 0x0130) return 'i';     // handled as a plain i (the dot-above combining mark is dropped)
    if (c == 0x0178) return 0xFF;
    if ((c >= 0x0139 && c <= 0x0148) || (c >= 0x0179 && c <= 0x017E)) return (c & 1) ? c + 1 : c;
    if (c == 0x0138 || c == 0x0149 || c == 0x017F) return c;
    return (c & 1) ? c : c + 1;
  }
  if (c >= 0x0391 && c <= 0x03AB && c != 0x03A2) return c + 32;
  if (c >= 0x0410 && c <= 0x042F) return c + 32;
  if (c >= 0x0400 && c <= 0x040F) return c + 80;
  if (c >= 0xFF21 && c <= 0xFF3A) return c + 32;
  return c;
}

// normalise: whitespace runs -> ' ', lowercase (leading / trailing spaces stay, as in HF's Replace)
std::string normalise(const std::string& in) {
  std::string out;
  bool in_ws = false;
  for (size_t i = 0; i < in.size();) {
    const uint32_t c = next_cp(in, i);
    if (is_space(c)) {
      if (!in_ws) out += ' ';
      in_ws = true;
    } else {
      in_ws = false;
      put_utf8(out, lower(c));
    }
  }
  return out;
}

bool starts_with(const std::string& s, size_t i, const char* lit) {
  const size_t n = strlen(lit);
  return s.compare(i, n, lit) == 0;
}

// the Split regex, leftmost match with the alternatives tried in order; spaces between matches vanish
void pre_tokenize(const std::string& s, std::vector<std::string>& words) {
  static const char* kSpecial[] = {"<|startoftext|>", "<|endoftext|>"};
  static const char* kContr[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
  size_t i = 0;
  while (i < s.size()) {
    bool done = false;
    for (const char* sp : kSpecial)
      if (starts_with(s, i, sp)) { words.emplace_back(sp); i += strlen(sp); done = true; break; }
    if (done) continue;
    for (const char* ct : kContr)
      if (starts_with(s, i, ct)) { words.emplace_back(ct); i += strlen(ct); done = true; break; }
    if (done) continue;
    size_t j = i;
    const uint32_t c = next_cp(s, j);
    if (is_space(c)) { i = j; continue; }
    if (is_letter(c)) {
      size_t k = j;
      while (k < s.size()) { size_t t = k; if (!is_letter(next_cp(s, t))) break; k = t; }
      words.emplace_back(s, i, k - i);
      i = k;
    } else if (is_number(c)) {
      words.emplace_back(s, i, j - i);
      i = j;
    } else {
      size_t k = j;
      while (k < s.size()) {
        size_t t = k;
        const uint32_t d = next_cp(s, t);
        if (is_space(d) || is_letter(d) || is_number(d)) break;
        k = t;
      }
      words.emplace_back(s, i, k - i);
      i = k;
    }
  }
}

const std::vector<int32_t>& bpe(mcm_tokenizer* t, const std::string& word) {
  auto it = t->cache.find(word);
  if (it != t->cache.end()) return it->second;
  std::vector<std::string> sym;
  for (unsigned char b : word) sym.push_back(t->b2u[b]);
  if (!sym.empty()) sym.back() += "</w>";
  while (sym.size() > 1) {
    int best = INT32_MAX;
    for (size_t i = 0; i + 1 < sym.size(); ++i) {
      auto r = t->rank.find(sym[i] + " " + sym[i + 1]);
      if (r != t->rank.end() && r->second < best) best = r->second;
    }
    if (best == INT32_MAX) break;
    std::vector<std::string> next;
    for (size_t i = 0; i < sym.size();) {  // merge every occurrence of the best pair, left to right
      if (i + 1 < sym.size()) {
        auto r = t->rank.find(sym[i] + " " + sym[i + 1]);
        if (r != t->rank.end() && r->second == best) { next.push_back(sym[i] + sym[i + 1]); i += 2; continue; }
      }
      next.push_back(sym[i]);
      ++i;
    }
    sym.swap(next);
  }
  std::vector<int32_t> ids;
  for (auto& s : sym) {
    auto v = t->vocab.find(s);
    ids.push_back(v != t->vocab.end() ? v->second : t->unk);
  }
  return t->cache.emplace(word, std::move(ids)).first->second;
}

bool read_file(const char* path, std::string& out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
  fclose(f);
  return true;
}

// {"token": id, ...} with JSON string escapes; nothing else is expected in vocab.json
bool parse_vocab(const std::string& js, std::unordered_map<std::string, int32_t>& vocab) {
  size_t i = 0;
  auto ws = [&] { while (i < js.size() && (js[i] == ' ' || js[i] == '\n' || js[i] == '\r' || js[i] == '\t')) ++i; };
  ws();
  if (i >= js.size() || js[i] != '{') return false;
  ++i;
  for (;;) {
    ws();
    if (i < js.size() && js[i] == '}') return true;
    if (i >= js.size() || js[i] != '"') return false;
    ++i;
    std::string key;
    while (i < js.size() && js[i] != '"') {
      if (js[i] == '\\' && i + 1 < js.size()) {
        const char e = js[i + 1];
        i += 2;
        if (e == 'u' && i + 4 <= js.size()) {
          uint32_t cp = (uint32_t)strtoul(js.substr(i, 4).c_str(), nullptr, 16);
          i += 4;
          if (cp >= 0xD800 && cp <= 0xDBFF && i + 6 <= js.size() && js[i] == '\\' && js[i + 1] == 'u') {
            const uint32_t lo = (uint32_t)strtoul(js.substr(i + 2, 4).c_str(), nullptr, 16);
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            i += 6;
          }
          put_utf8(key, cp);
        } else {
          key += e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e == 'b' ? '\b' : e == 'f' ? '\f' : e;
        }
      } else {
        key += js[i++];
      }
    }
    if (i >= js.size()) return false;
    ++i;
    ws();
    if (i >= js.size() || js[i] != ':') return false;
    ++i;
    ws();
    char* end = nullptr;
    const long v = strtol(js.c_str() + i, &end, 10);
    if (end == js.c_str() + i) return false;
    i = (size_t)(end - js.c_str());
    vocab[key] = (int32_t)v;
    ws();
    if (i < js.size() && js[i] == ',') { ++i; continue; }
    if (i < js.size() && js[i] == '}') return true;
    return false;
  }
}

thread_local std::string g_tok_err;

}  // namespace

extern "C" {

int mcm_tokenizer_create(const char* vocab_json_path, const char* merges_txt_path, mcm_tokenizer** out) {
  if (!vocab_json_path || !merges_txt_path || !out) return MCM_EINVAL;
  auto* t = new mcm_tokenizer;
  std::string js, mg;
  if (!read_file(vocab_json_path, js) || !parse_vocab(js, t->vocab)) {
    g_tok_err = std::string("cannot read / parse ") + vocab_json_path;
    delete t;
    return MCM_EINVAL;
  }
  if (!read_file(merges_txt_path, mg)) {
    g_tok_err = std::string("cannot read ") + merges_txt_path;
    delete t;
    return MCM_EINVAL;
  }
  int32_t r = 0;
  size_t pos = 0;
  bool first = true;
  while (pos < mg.size()) {
    size_t nl = mg.find('\n', pos);
    if (nl == std::string::npos) nl = mg.size();
    std::string line = mg.substr(pos, nl - pos);
    pos = nl + 1;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (first && line.compare(0, 8, "#version") == 0) { first = false; continue; }
    first = false;
    if (line.empty()) continue;
    if (line.find(' ') == std::string::npos) continue;
    t->rank.emplace(line, r++);
  }
  // GPT-2 bytes_to_unicode: printable bytes map to themselves, the rest to U+0100 + n
  int n = 0;
  for (int b = 0; b < 256; ++b) {
    const bool keep = (b >= '!' && b <= '~') || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
    put_utf8(t->b2u[b], keep ? (uint32_t)b : (uint32_t)(256 + n++));
  }
  auto id = [&](const char* s) { auto it = t->vocab.find(s); return it == t->vocab.end() ? -1 : it->second; };
  t->bos = id("<|startoftext|>");
  t->eos = id("<|endoftext|>");
  t->unk = t->eos;
  t->pad = t->eos;
  if (t->bos < 0 || t->eos < 0) {
    g_tok_err = "vocabulary lacks <|startoftext|> / <|endoftext|>";
    delete t;
    return MCM_EINVAL;
  }
  *out = t;
  return MCM_OK;
}

void mcm_tokenizer_destroy(mcm_tokenizer* t) { delete t; }

const char* mcm_tokenizer_last_error(const mcm_tokenizer* t) { return t ? t->err.c_str() : g_tok_err.c_str(); }

int32_t mcm_tokenizer_vocab_size(const mcm_tokenizer* t) { return t ? (int32_t)t->vocab.size() : 0; }

int mcm_tokenizer_encode(mcm_tokenizer* t, const char* const* texts, int32_t n, int32_t capacity,
                         int32_t* ids_out, int32_t* mask_out, int32_t* seq_len_out) {
  if (!t) return MCM_EINVAL;
  if (!texts || n <= 0 || capacity <= 0 || !ids_out || !seq_len_out) {
    t->err = "bad argument";
    return MCM_EINVAL;
  }
  std::vector<std::vector<int32_t>> rows((size_t)n);
  int32_t longest = 0;
  for (int32_t k = 0; k < n; ++k) {
    if (!texts[k]) { t->err = "null text"; return MCM_EINVAL; }
    std::vector<std::string> words;
    pre_tokenize(normalise(texts[k]), words);
    auto& row = rows[(size_t)k];
    row.push_back(t->bos);
    for (auto& w : words) {
      if (w == "<|startoftext|>") { row.push_back(t->bos); continue; }
      if (w == "<|endoftext|>") { row.push_back(t->eos); continue; }
      const auto& ids = bpe(t, w);
      row.insert(row.end(), ids.begin(), ids.end());
    }
    row.push_back(t->eos);
    if ((int32_t)row.size() > longest) longest = (int32_t)row.size();
  }
  *seq_len_out = longest;
  if (longest > capacity) {
    t->err = "a prompt is longer than the output capacity";
    return MCM_ERANGE;
  }
  for (int32_t k = 0; k < n; ++k) {
    const auto& row = rows[(size_t)k];
    for (int32_t j = 0; j < longest; ++j) {
      const bool real = j < (int32_t)row.size();
      ids_out[(size_t)k * longest + j] = real ? row[(size_t)j] : t->pad;
      if (mask_out) mask_out[(size_t)k * longest + j] = real ? 1 : 0;
    }
  }
  return MCM_OK;
}

}  // extern "C"
