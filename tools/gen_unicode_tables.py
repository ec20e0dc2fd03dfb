"""gen_unicode_tables.py — writes mcm_amd/csrc/unicode_tables.inc, the Unicode data of the C++ CLIP tokenizer
(mcm_amd/csrc/tokenizer.cpp).

What the tables must reproduce is the text pipeline HF transformers' CLIPTokenizer configures (third-party; the
tokenizer the reference calls at utils/detection_util.py:216,228): NFC -> whitespace runs to one space -> lowercase ->
split on  's|'t|'re|'ve|'m|'ll|'d | \\p{L}+ | \\p{N} | [^\\s\\p{L}\\p{N}]+ .  Sources, all local to the build container:

  letter / number / space classes   probed code point by code point from the installed `tokenizers` backend's own
                                    Split pre-tokenizer (the classes are Unicode properties; the probe pins the Unicode
                                    VERSION to the one the checker in tests/test_tokenizer_bpe.py uses)
  lowercase                         probed from the backend's Lowercase normalizer (per character, no final-sigma context)
  NFC                               probed from the backend's NFD / NFC normalizers (decompositions, combining classes by
                                    ordering probes, primary composites): its normalisation data is an older Unicode
                                    version than its regex classes, and the checker is what has to be matched.  Python's
                                    unicodedata only supplies the candidates (marks, one-level decompositions).  Hangul is
                                    algorithmic in the C++.

Run here (needs `tokenizers`); the output is data and is committed.  python tools/gen_unicode_tables.py"""
import os
import sys
import unicodedata

from tokenizers import Regex, normalizers, pre_tokenizers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mcm_amd", "csrc", "unicode_tables.inc")
PATTERN = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"


def code_points():
    for c in range(0x110000):
        if not 0xD800 <= c <= 0xDFFF:
            yield c


def ranges(cps):
    out, lo, prev = [], None, None
    for c in cps:
        if lo is None:
            lo = prev = c
        elif c == prev + 1:
            prev = c
        else:
            out.append((lo, prev))
            lo = prev = c
    if lo is not None:
        out.append((lo, prev))
    return out


def main():
    split = pre_tokenizers.Split(Regex(PATTERN), behavior="removed", invert=True)
    lower = normalizers.Lowercase()
    nfc_hf = normalizers.NFC()
    letters, numbers, spaces, lowers = [], [], [], []
    for c in code_points():
        ch = chr(c)
        n1 = len(split.pre_tokenize_str("a" + ch + "a"))
        if n1 == 1:
            letters.append(c)
        elif n1 == 2:
            spaces.append(c)
        elif len(split.pre_tokenize_str(ch + ch)) == 2:
            numbers.append(c)
        lo = lower.normalize_str(ch)
        if lo != ch:
            lowers.append((c, [ord(x) for x in lo]))
    # NFC data, as the backend's NFC has it (its normalisation tables are an OLDER Unicode version than its regex
    # classes: marks added later have class 0 there and are not reordered).  Decompositions: its NFD of the single code
    # point.  Combining classes: ordering probes against one reference mark per class value — NFD("a" + r + m) swaps the
    # two marks iff 0 < ccc(m) < ccc(r).  Primary composites: pairs (from unicodedata's one-level decompositions) that its
    # NFC composes back.
    nfd_hf = normalizers.NFD()
    by_class = {}
    for c in code_points():
        k = unicodedata.combining(chr(c))
        if k:
            by_class.setdefault(k, []).append(c)
    values = sorted(by_class)

    def swaps(r, m):
        return r != m and nfd_hf.normalize_str("a" + chr(r) + chr(m)) == "a" + chr(m) + chr(r)

    refs = {}
    for v in values:  # a reference per value that the backend orders consistently against the references below it
        for cand in by_class[v]:
            if all(swaps(cand, refs[u]) and not swaps(refs[u], cand) for u in refs):
                refs[v] = cand
                break
        else:
            print(f"combining class {v}: no mark of it is ordered by the backend (added after its Unicode version)", file=sys.stderr)
    values = sorted(refs)
    decomp, ccc, comp = [], [], []
    n_dropped = 0
    for v in sorted(by_class):
        for m in by_class[v]:
            below = [u for u in values if not swaps(refs[u], m)]  # classes u <= ccc(m); every u when ccc(m) == 0
            k = max(below) if below else 0
            if len(below) == len(values) and not swaps(m, refs[values[0]]):
                k = 0  # a starter for the backend: the lowest-class mark after it stays where it is
            if k != v:
                n_dropped += 1
            if k:
                ccc.append((m, k))
    ccc.sort()
    nfc_diff = 0
    for c in code_points():
        ch = chr(c)
        if 0xAC00 <= c <= 0xD7A3:
            continue  # Hangul syllables: algorithmic
        full = nfd_hf.normalize_str(ch)
        if full != ch:
            decomp.append((c, [ord(x) for x in full]))
            d = unicodedata.decomposition(ch)
            if d and not d.startswith("<"):
                one = [int(x, 16) for x in d.split()]
                if len(one) == 2 and nfc_hf.normalize_str("".join(map(chr, one))) == ch:
                    comp.append((one[0], one[1], c))
        if unicodedata.category(ch) != "Cn" and nfc_hf.normalize_str(ch) != unicodedata.normalize("NFC", ch):
            nfc_diff += 1
    print(f"marks whose class differs from Unicode {unicodedata.unidata_version}'s in the backend: {n_dropped}", file=sys.stderr)
    comp.sort()
    print(f"letters {len(letters)} cps / {len(ranges(letters))} ranges, numbers {len(numbers)} / {len(ranges(numbers))}, "
          f"spaces {spaces}, lowercase {len(lowers)}, decompositions {len(decomp)}, ccc {len(ccc)}, composites {len(comp)}; "
          f"single code points (assigned in Unicode {unicodedata.unidata_version}) whose NFC differs between unicodedata and "
          f"the backend: {nfc_diff}", file=sys.stderr)

    def emit_ranges(f, name, rs):
        f.write(f"static const uint32_t {name}[][2] = {{\n")
        for i in range(0, len(rs), 6):
            f.write("  " + " ".join(f"{{0x{a:X},0x{b:X}}}," for a, b in rs[i:i + 6]) + "\n")
        f.write("};\n")

    with open(OUT, "w") as f:
        f.write("// unicode_tables.inc — GENERATED by tools/gen_unicode_tables.py (data only; see its header for the sources).\n")
        f.write("// Classes, lowercase and NFC data as the installed tokenizers backend has them (probed; candidates from Python unicodedata).\n")
        emit_ranges(f, "kLetterRanges", ranges(letters))
        emit_ranges(f, "kNumberRanges", ranges(numbers))
        emit_ranges(f, "kSpaceRanges", ranges(spaces))
        f.write("// code point -> lowercase sequence (0-terminated, at most 3)\n")
        f.write("static const uint32_t kLower[][4] = {\n")
        for i in range(0, len(lowers), 4):
            f.write("  " + " ".join("{" + ",".join(f"0x{x:X}" for x in [c] + (s + [0, 0, 0])[:3]) + "}," for c, s in lowers[i:i + 4]) + "\n")
        f.write("};\n")
        assert max(len(s) for _, s in lowers) <= 3
        f.write("// code point -> full canonical decomposition (0-terminated, at most 4)\n")
        assert max(len(s) for _, s in decomp) <= 4
        f.write("static const uint32_t kDecomp[][5] = {\n")
        for i in range(0, len(decomp), 4):
            f.write("  " + " ".join("{" + ",".join(f"0x{x:X}" for x in [c] + (s + [0, 0, 0, 0])[:4]) + "}," for c, s in decomp[i:i + 4]) + "\n")
        f.write("};\n")
        f.write("// canonical combining class (non-zero only)\n")
        f.write("static const uint32_t kCcc[][2] = {\n")
        for i in range(0, len(ccc), 8):
            f.write("  " + " ".join(f"{{0x{c:X},{k}}}," for c, k in ccc[i:i + 8]) + "\n")
        f.write("};\n")
        f.write("// primary composites: (first, second) -> composed, sorted by (first, second)\n")
        f.write("static const uint32_t kComp[][3] = {\n")
        for i in range(0, len(comp), 5):
            f.write("  " + " ".join(f"{{0x{a:X},0x{b:X},0x{c:X}}}," for a, b, c in comp[i:i + 5]) + "\n")
        f.write("};\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
