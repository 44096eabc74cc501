// regex_nfa.hpp -- the `=~ /pattern/` conditions of the -F query language (RegexpFieldFilter / RegexpTagFilter,
// sambamba/utils/common/filtering.d:299-345: `!match(text, regex).empty`, i.e. an unanchored search).
//
// The pattern is compiled on the host into a Thompson NFA of at most 64 states (so that a set of states is one
// 64-bit mask) and simulated per record on the device -- the same simulation code runs on the host for reference
// names and for the CPU tests.  Syntax: the common core of D's std.regex / ECMAScript: literals, `.`, classes
// `[a-z]` `[^...]`, escapes `\d \w \s \D \W \S` and escaped metacharacters, anchors `^` `$`, groups `( )` `(?: )`,
// alternation `|`, quantifiers `* + ? {m} {m,} {m,n}` (lazy forms accepted: a search only asks whether a match exists).
// Back-references, look-around and the g/x/U/m/s options are outside the subset (SBX_EUNSUPPORTED); option `i` is supported.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sbx_depth.h"

#if defined(__HIPCC__)
#define SBX_HD __host__ __device__
#else
#define SBX_HD
#endif

namespace sbx {

enum : uint8_t { RE_CHAR = 0, RE_ANY = 1, RE_CLASS = 2, RE_SPLIT = 3, RE_JMP = 4, RE_BOL = 5, RE_EOL = 6, RE_MATCH = 7 };

// epsilon closure of a state set at a given position (conditional on begin / end of the text)
SBX_HD inline uint64_t re_closure(const sbx_regex& re, uint64_t set, bool at_begin, bool at_end) {
    uint64_t done = 0, work = set;
    while (work) {
        const int s = __builtin_ctzll(work);
        work &= work - 1;
        if ((done >> s) & 1ull) continue;
        done |= 1ull << s;
        const uint8_t ty = re.states[s].type, a = re.states[s].a, b = re.states[s].b;
        uint64_t add = 0;
        if (ty == RE_SPLIT) add = (1ull << a) | (1ull << b);
        else if (ty == RE_JMP) add = 1ull << a;
        else if (ty == RE_BOL) add = at_begin ? (1ull << a) : 0;
        else if (ty == RE_EOL) add = at_end ? (1ull << a) : 0;
        work |= add & ~done;
    }
    return done;
}

// one character of the text
SBX_HD inline uint64_t re_step(const sbx_regex& re, uint64_t set, uint8_t ch) {
    uint64_t nxt = 0;
    while (set) {
        const int s = __builtin_ctzll(set);
        set &= set - 1;
        const uint8_t ty = re.states[s].type;
        bool ok = false;
        if (ty == RE_CHAR) ok = re.states[s].c == ch;
        else if (ty == RE_ANY) ok = ch != '\n' && ch != '\r';
        else if (ty == RE_CLASS) ok = (re.classes[re.states[s].c][ch >> 3] >> (ch & 7)) & 1;
        if (ok) nxt |= 1ull << re.states[s].a;
    }
    return nxt;
}

// does the pattern match anywhere in text[0, n)?  `at(i)` yields character i.
template <class At>
SBX_HD inline bool re_search(const sbx_regex& re, uint32_t n, At at) {
    if (re.n_states == 0) return false;
    uint64_t match_bit = 0;
    for (int s = 0; s < re.n_states; ++s) if (re.states[s].type == RE_MATCH) match_bit |= 1ull << s;
    uint64_t cur = 0;
    for (uint32_t pos = 0;; ++pos) {
        cur = re_closure(re, cur | (1ull << re.start), pos == 0, pos == n);
        if (cur & match_bit) return true;
        if (pos == n) return false;
        cur = re_step(re, cur, at(pos));
    }
}

// ---- host: pattern -> NFA ----------------------------------------------------------------------------------------
class RegexCompiler {
public:
    RegexCompiler(const std::string& pattern, bool icase, sbx_regex* out) : p_(pattern), icase_(icase), re_(out) {
        memset(re_, 0, sizeof *re_);
    }
    void compile() {
        Frag f = alternation();
        if (i_ != p_.size()) fail("unbalanced ')'");
        const int m = state(RE_MATCH, 0, 0, 0);
        patch(f, m);
        re_->start = (uint8_t)f.start;
    }

private:
    struct Frag { int start; std::vector<std::pair<int, int>> out; };   // dangling arrows: (state, which: 0 = a, 1 = b)
    std::string p_;
    size_t i_ = 0;
    bool icase_;
    sbx_regex* re_;
    [[noreturn]] void fail(const std::string& why) { throw Error(SBX_EUNSUPPORTED, "filter: regular expression /" + p_ + "/: " + why); }
    int state(uint8_t type, int a, int b, int c) {
        if (re_->n_states >= SBX_REGEX_STATES) fail("too complex for the device (more than 64 NFA states)");
        auto& s = re_->states[re_->n_states];
        s.type = type; s.a = (uint8_t)a; s.b = (uint8_t)b; s.c = (uint8_t)c;
        return re_->n_states++;
    }
    void patch(const Frag& f, int to) {
        for (auto& o : f.out) { if (o.second) re_->states[o.first].b = (uint8_t)to; else re_->states[o.first].a = (uint8_t)to; }
    }
    int new_class(const uint8_t bits[32]) {
        for (int k = 0; k < re_->n_classes; ++k) if (memcmp(re_->classes[k], bits, 32) == 0) return k;
        if (re_->n_classes >= SBX_REGEX_CLASSES) fail("too many character classes for the device");
        memcpy(re_->classes[re_->n_classes], bits, 32);
        return re_->n_classes++;
    }
    static void set_bit(uint8_t bits[32], int ch) { bits[(ch & 255) >> 3] |= (uint8_t)(1u << (ch & 7)); }
    void add_char(uint8_t bits[32], int ch) {
        set_bit(bits, ch);
        if (icase_) { if (ch >= 'a' && ch <= 'z') set_bit(bits, ch - 32); if (ch >= 'A' && ch <= 'Z') set_bit(bits, ch + 32); }
    }
    void add_escape_class(uint8_t bits[32], char e) {
        uint8_t t[32] = {0};
        const char lower = (char)(e | 32);
        for (int ch = 0; ch < 256; ++ch) {
            bool in = lower == 'd' ? (ch >= '0' && ch <= '9')
                    : lower == 'w' ? ((ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_')
                    : (ch == ' ' || (ch >= 9 && ch <= 13));
            if (e >= 'A' && e <= 'Z') in = !in && ch < 128;
            if (in) set_bit(t, ch);
        }
        for (int k = 0; k < 32; ++k) bits[k] |= t[k];
    }
    static int escaped_literal(char e) {
        switch (e) { case 'n': return '\n'; case 't': return '\t'; case 'r': return '\r'; case 'f': return '\f'; case 'v': return '\v'; case '0': return 0; default: return (uint8_t)e; }
    }
    Frag single(uint8_t type, int c) {
        const int s = state(type, 0, 0, c);
        return Frag{s, {{s, 0}}};
    }
    Frag char_frag(int ch) {
        if (icase_ && ((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'))) {
            uint8_t bits[32] = {0};
            add_char(bits, ch);
            return single(RE_CLASS, new_class(bits));
        }
        return single(RE_CHAR, ch);
    }
    Frag atom() {
        if (i_ >= p_.size()) fail("unexpected end");
        const char ch = p_[i_++];
        if (ch == '(') {
            if (p_.compare(i_, 2, "?:") == 0) i_ += 2;
            else if (i_ < p_.size() && p_[i_] == '?') fail("look-around / named groups are not supported");
            Frag f = alternation();
            if (i_ >= p_.size() || p_[i_] != ')') fail("missing ')'");
            ++i_;
            return f;
        }
        if (ch == '.') return single(RE_ANY, 0);
        if (ch == '^') return single(RE_BOL, 0);
        if (ch == '$') return single(RE_EOL, 0);
        if (ch == '[') {
            uint8_t bits[32] = {0};
            bool neg = false;
            if (i_ < p_.size() && p_[i_] == '^') { neg = true; ++i_; }
            bool first = true;
            for (;; first = false) {
                if (i_ >= p_.size()) fail("missing ']'");
                char c = p_[i_++];
                if (c == ']' && !first) break;
                int lo;
                if (c == '\\') {
                    if (i_ >= p_.size()) fail("dangling backslash");
                    const char e = p_[i_++];
                    if (strchr("dwsDWS", e)) { add_escape_class(bits, e); continue; }
                    lo = escaped_literal(e);
                } else lo = (uint8_t)c;
                if (i_ + 1 < p_.size() && p_[i_] == '-' && p_[i_ + 1] != ']') {
                    ++i_;
                    int hi = (uint8_t)p_[i_++];
                    if (hi == '\\') { if (i_ >= p_.size()) fail("dangling backslash"); hi = escaped_literal(p_[i_++]); }
                    if (hi < lo) fail("inverted range in a character class");
                    for (int x = lo; x <= hi; ++x) add_char(bits, x);
                } else add_char(bits, lo);
            }
            if (neg) for (int k = 0; k < 32; ++k) bits[k] = (uint8_t)~bits[k];
            return single(RE_CLASS, new_class(bits));
        }
        if (ch == '\\') {
            if (i_ >= p_.size()) fail("dangling backslash");
            const char e = p_[i_++];
            if (strchr("dwsDWS", e)) { uint8_t bits[32] = {0}; add_escape_class(bits, e); return single(RE_CLASS, new_class(bits)); }
            if (e >= '1' && e <= '9') fail("back-references are not supported");
            if (e == 'b' || e == 'B') fail("word boundaries are not supported");
            return char_frag(escaped_literal(e));
        }
        if (ch == '*' || ch == '+' || ch == '?' || ch == '{') fail("nothing to repeat");
        return char_frag((uint8_t)ch);
    }
    // a copy of the pattern text [a, b) compiled again (counted repetition)
    Frag again(size_t a, size_t b) {
        const size_t save = i_;
        i_ = a;
        Frag f = atom();
        if (i_ != b) fail("internal: repeat bounds");
        i_ = save;
        return f;
    }
    Frag star(Frag f) {
        const int s = state(RE_SPLIT, f.start, 0, 0);
        patch(f, s);
        return Frag{s, {{s, 1}}};
    }
    Frag quest(Frag f) {
        const int s = state(RE_SPLIT, f.start, 0, 0);
        f.out.push_back({s, 1});
        return Frag{s, f.out};
    }
    Frag cat(Frag a, const Frag& b) { patch(a, b.start); return Frag{a.start, b.out}; }
    Frag repeat() {
        const size_t a0 = i_;
        Frag f = atom();
        const size_t a1 = i_;
        while (i_ < p_.size()) {
            const char q = p_[i_];
            if (q == '*') { ++i_; f = star(f); }
            else if (q == '+') { ++i_; Frag g = star(again(a0, a1)); f = cat(f, g); }
            else if (q == '?') { ++i_; f = quest(f); }
            else if (q == '{') {
                size_t j = i_ + 1;
                auto num = [&](int* v) { if (j >= p_.size() || !isdigit((unsigned char)p_[j])) return false; *v = 0; while (j < p_.size() && isdigit((unsigned char)p_[j])) *v = *v * 10 + (p_[j++] - '0'); return true; };
                int lo = 0, hi = -1;
                if (!num(&lo)) break;                       // a literal '{'
                if (j < p_.size() && p_[j] == ',') { ++j; int h; if (num(&h)) hi = h; else hi = -2; } else hi = lo;
                if (j >= p_.size() || p_[j] != '}') break;
                i_ = j + 1;
                if (hi >= 0 && hi < lo) fail("inverted repeat count");
                if (lo > 32 || hi > 32) fail("repeat counts above 32 are not supported");
                // x{lo,hi} = x ... x (lo times) followed by (x?){hi - lo} or x* when unbounded
                Frag r{-1, {}};
                bool have = false;
                auto append = [&](Frag g) { if (!have) { r = g; have = true; } else r = cat(r, g); };
                for (int k = 0; k < lo; ++k) append(k == 0 ? f : again(a0, a1));
                if (lo == 0) {          // the already compiled copy becomes optional / starred
                    if (hi == -2) append(star(f));
                    else { if (hi == 0) fail("{0} is not supported"); append(quest(f)); for (int k = 1; k < hi; ++k) append(quest(again(a0, a1))); }
                } else {
                    if (hi == -2) append(star(again(a0, a1)));
                    else for (int k = lo; k < hi; ++k) append(quest(again(a0, a1)));
                }
                f = r;
            } else break;
            if (i_ < p_.size() && p_[i_] == '?') ++i_;      // lazy quantifier: same language
        }
        return f;
    }
    Frag sequence() {
        Frag f{-1, {}};
        bool have = false;
        while (i_ < p_.size() && p_[i_] != '|' && p_[i_] != ')') {
            Frag g = repeat();
            if (!have) { f = g; have = true; } else f = cat(f, g);
        }
        if (!have) { const int s = state(RE_JMP, 0, 0, 0); f = Frag{s, {{s, 0}}}; }   // empty branch
        return f;
    }
    Frag alternation() {
        Frag f = sequence();
        while (i_ < p_.size() && p_[i_] == '|') {
            ++i_;
            Frag g = sequence();
            const int s = state(RE_SPLIT, f.start, g.start, 0);
            f.out.insert(f.out.end(), g.out.begin(), g.out.end());
            f = Frag{s, f.out};
        }
        return f;
    }
};

}  // namespace sbx
