// fsk_fasta.cpp — native FASTA-like tokeniser behind fsk_read_fasta (include/fastsk_amd.h).
//
// Replaces, for large inputs, the per-character Python loop of the reference's reader
// (FastaUtility.read_data + Vocabulary.add, src/fastsk/utils.py:5-96): alternating ">label" /
// sequence lines, every line stripped of surrounding white space and lower-cased (utils.py:78),
// labels restricted to {-1, 0, 1} (utils.py:84-85), token ids handed out in first-seen order
// starting at 1 (id 0 is reserved, utils.py:13) from a table the caller keeps between files, so
// that train and test files share ids (one Vocabulary per FastaUtility, utils.py:39-48).
// Host code only: no HIP call in this file.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fastsk_amd.h"

namespace {

// what Python's str.strip() removes, restricted to ASCII (non-ASCII input is refused)
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f); }

void put_error(char* err, int32_t cap, const std::string& msg) {
    if (!err || cap <= 0) return;
    snprintf(err, (size_t)cap, "%s", msg.c_str());
}

// int(text) for the label: optional white space, optional sign, decimal digits (underscores between
// digits as Python allows), optional white space
bool parse_label(const unsigned char* p, const unsigned char* q, long* out) {
    while (p < q && is_space(*p)) ++p;
    while (q > p && is_space(q[-1])) --q;
    if (p == q) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
    if (p == q || *p < '0' || *p > '9') return false;
    long v = 0;
    bool last_digit = false;
    for (; p < q; ++p) {
        if (*p >= '0' && *p <= '9') {
            if (v < 1000000) v = v * 10 + (*p - '0');
            last_digit = true;
        } else if (*p == '_' && last_digit) {
            last_digit = false;
        } else {
            return false;
        }
    }
    if (!last_digit) return false;
    *out = neg ? -v : v;
    return true;
}

}  // namespace

extern "C" int fsk_read_fasta(const char* path, int32_t* vocab256, int32_t* next_id, int32_t* tokens, int64_t tokens_cap,
                              int64_t* offsets, int32_t* labels, int64_t seq_cap, int64_t* n_seq, int64_t* n_tokens, char* err,
                              int32_t err_cap) {
    if (!path || !vocab256 || !next_id || !n_seq || !n_tokens) {
        put_error(err, err_cap, "null argument");
        return FSK_EINVAL;
    }
    *n_seq = *n_tokens = 0;
    FILE* f = fopen(path, "rb");
    if (!f) {
        put_error(err, err_cap, std::string("cannot open ") + path);
        return FSK_EINVAL;
    }
    std::vector<unsigned char> buf;
    {
        unsigned char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
        fclose(f);
    }
    const bool fill = tokens != nullptr || offsets != nullptr || labels != nullptr;
    if (fill && (!offsets || !labels || (tokens_cap > 0 && !tokens))) {
        put_error(err, err_cap, "tokens, offsets and labels must be given together");
        return FSK_EINVAL;
    }
    int64_t ns = 0, nt = 0;
    bool expect_label = true;
    size_t line_no = 0;
    const unsigned char* p = buf.data();
    const unsigned char* end = p + buf.size();
    if (fill && seq_cap >= 0 && offsets) offsets[0] = 0;
    while (p < end) {
        // one line: up to "\n", "\r\n" or a lone "\r" (universal newlines, as Python's text mode)
        const unsigned char* q = p;
        while (q < end && *q != '\n' && *q != '\r') ++q;
        const unsigned char* next = q;
        if (next < end) next += (*next == '\r' && next + 1 < end && next[1] == '\n') ? 2 : 1;
        ++line_no;
        const unsigned char* a = p;
        const unsigned char* b = q;
        while (a < b && is_space(*a)) ++a;
        while (b > a && is_space(b[-1])) --b;
        for (const unsigned char* c = a; c < b; ++c)
            if (*c >= 128) {
                put_error(err, err_cap, "non-ASCII byte on line " + std::to_string(line_no));
                return FSK_EUNSUPPORTED;  // the caller falls back to its text-mode reader
            }
        if (expect_label) {
            // exactly one '>' on the line; the label is what follows it (utils.py:79-81)
            const unsigned char* gt = nullptr;
            int n_gt = 0;
            for (const unsigned char* c = a; c < b; ++c)
                if (*c == '>') { if (!n_gt) gt = c; ++n_gt; }
            long lab = 0;
            if (n_gt != 1 || !parse_label(gt + 1, b, &lab) || lab < -1 || lab > 1) {
                put_error(err, err_cap, "line " + std::to_string(line_no) + ": expected a label line '>-1', '>0' or '>1'");
                return FSK_EINVAL;
            }
            if (fill) {
                if (ns >= seq_cap) { put_error(err, err_cap, "sequence capacity too small"); return FSK_EINVAL; }
                labels[ns] = (int32_t)lab;
            }
        } else {
            for (const unsigned char* c = a; c < b; ++c) {
                unsigned char ch = *c;
                if (ch >= 'A' && ch <= 'Z') ch = (unsigned char)(ch - 'A' + 'a');
                int32_t id = vocab256[ch];
                if (id <= 0) id = vocab256[ch] = (*next_id)++;
                if (fill) {
                    if (nt >= tokens_cap) { put_error(err, err_cap, "token capacity too small"); return FSK_EINVAL; }
                    tokens[nt] = id;
                }
                ++nt;
            }
            if (fill) offsets[ns + 1] = nt;
            ++ns;
        }
        expect_label = !expect_label;
        p = next;
    }
    if (!expect_label) {  // a label without its sequence: `assert len(X) == len(Y)`, utils.py:94
        put_error(err, err_cap, "the file ends after a label line");
        return FSK_EINVAL;
    }
    *n_seq = ns;
    *n_tokens = nt;
    return FSK_OK;
}
