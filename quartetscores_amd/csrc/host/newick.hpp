// newick.hpp -- minimal tree model + Newick reader/writer of the C++ host.
//
// Stands where genesis (DefaultTreeNewickReader / NewickInputIterator / NewickWriter) stands in
// the reference (QuartetScores.cpp:23-32,97-98,149-158; QuartetCounterLookup.hpp:202-206). Only
// what the hot path needs: topology, leaf names, verbatim branch lengths, edge comments on output.
// Dialect: nested parentheses, labels plain or 'quoted', ':length', [comments] skipped, ';' ends a tree.
#pragma once

#include <cstdint>
#include <functional>
#include <istream>
#include <stdexcept>
#include <string>
#include <vector>

namespace qsh {

struct Tree {
    // nodes in preorder; node 0 is the root
    std::vector<int32_t> parent;
    std::vector<std::vector<int32_t>> children;
    std::vector<std::string> name;
    std::vector<std::string> length; // verbatim text after ':' ("" = none)

    size_t node_count() const { return parent.size(); }
    size_t edge_count() const { return parent.empty() ? 0 : parent.size() - 1; }
    bool is_leaf(size_t v) const { return children[v].empty(); }
    size_t leaf_count() const {
        size_t k = 0;
        for (size_t v = 0; v < parent.size(); ++v) k += children[v].empty();
        return k;
    }
};

struct NewickError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Streaming reader: one tree per call, from a text buffer.
class NewickReader {
public:
    explicit NewickReader(const std::string &text) : s_(text) {}
    // returns false at end of input
    bool next(Tree &out) {
        skip();
        if (i_ >= s_.size()) return false;
        parse_one(out);
        return true;
    }
    size_t offset() const { return i_; }

private:
    const std::string &s_;
    size_t i_ = 0;

    void skip() {
        while (i_ < s_.size()) {
            char c = s_[i_];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') ++i_;
            else if (c == '[') {
                size_t j = s_.find(']', i_);
                if (j == std::string::npos) throw NewickError("unterminated comment");
                i_ = j + 1;
            } else break;
        }
    }
    std::string label() {
        skip();
        std::string out;
        if (i_ < s_.size() && s_[i_] == '\'') {
            ++i_;
            while (i_ < s_.size()) {
                if (s_[i_] == '\'') {
                    if (i_ + 1 < s_.size() && s_[i_ + 1] == '\'') { out.push_back('\''); i_ += 2; continue; }
                    ++i_;
                    break;
                }
                out.push_back(s_[i_++]);
            }
            return out;
        }
        while (i_ < s_.size()) {
            char c = s_[i_];
            if (c == '(' || c == ')' || c == ',' || c == ':' || c == ';' || c == '[' || c == ' ' || c == '\t' || c == '\n' || c == '\r') break;
            out.push_back(c);
            ++i_;
        }
        return out;
    }
    int32_t add(Tree &t, int32_t par) {
        int32_t id = (int32_t)t.parent.size();
        t.parent.push_back(par);
        t.children.emplace_back();
        t.name.emplace_back();
        t.length.emplace_back();
        if (par >= 0) t.children[par].push_back(id);
        return id;
    }
    void parse_one(Tree &t) {
        t = Tree();
        int32_t cur = add(t, -1);
        skip();
        if (i_ < s_.size() && s_[i_] != '(') t.name[cur] = label();
        while (i_ < s_.size()) {
            skip();
            if (i_ >= s_.size()) break;
            char c = s_[i_];
            if (c == '(') {
                cur = add(t, cur);
                ++i_;
                skip();
                if (i_ < s_.size() && s_[i_] != '(' && s_[i_] != ',' && s_[i_] != ')') t.name[cur] = label();
            } else if (c == ',') {
                if (t.parent[cur] < 0) throw NewickError("',' outside parentheses");
                cur = add(t, t.parent[cur]);
                ++i_;
                skip();
                if (i_ < s_.size() && s_[i_] != '(' && s_[i_] != ',' && s_[i_] != ')') t.name[cur] = label();
            } else if (c == ')') {
                if (t.parent[cur] < 0) throw NewickError("unbalanced ')'");
                cur = t.parent[cur];
                ++i_;
                skip();
                if (i_ < s_.size() && s_[i_] != '(' && s_[i_] != ')' && s_[i_] != ',' && s_[i_] != ':' && s_[i_] != ';')
                    t.name[cur] = label();
            } else if (c == ':') {
                ++i_;
                skip();
                size_t j = i_;
                while (j < s_.size() && s_[j] != '(' && s_[j] != ')' && s_[j] != ',' && s_[j] != ';' && s_[j] != '[' && s_[j] != ' ' && s_[j] != '\t' && s_[j] != '\n' && s_[j] != '\r') ++j;
                t.length[cur] = s_.substr(i_, j - i_);
                i_ = j;
            } else if (c == ';') {
                ++i_;
                break;
            } else {
                throw NewickError("unexpected character '" + std::string(1, c) + "' at offset " + std::to_string(i_));
            }
        }
        if (cur != 0) throw NewickError("unbalanced '('");
        // nodes were created in parse order, which is preorder, but a sibling is appended after the
        // whole previous subtree, so ids are already preorder-consistent
    }
};

inline std::string quote_label(const std::string &n) {
    bool need = false;
    for (char c : n) if (c == ' ' || c == '\t' || c == '(' || c == ')' || c == '[' || c == ']' || c == '\'' || c == ':' || c == ';' || c == ',') need = true;
    if (!need) return n;
    std::string o = "'";
    for (char c : n) { if (c == '\'') o += "''"; else o.push_back(c); }
    return o + "'";
}

// Element layout: name, ':length' (only if the input had one), '[comment]'. The reference's layout
// is decided by genesis v0.16.0's NewickWriter (absent here; parity unpinned).
inline std::string write_newick(const Tree &t, const std::function<std::string(size_t)> &comment) {
    std::string out;
    struct Frame { int32_t node; size_t next_child; };
    std::vector<Frame> st;
    st.push_back({0, 0});
    if (!t.children[0].empty()) out.push_back('(');
    while (!st.empty()) {
        Frame &f = st.back();
        const auto &ch = t.children[f.node];
        if (f.next_child < ch.size()) {
            if (f.next_child > 0) out.push_back(',');
            int32_t c = ch[f.next_child++];
            st.push_back({c, 0});
            if (!t.children[c].empty()) out.push_back('(');
            continue;
        }
        if (!ch.empty()) out.push_back(')');
        out += quote_label(t.name[f.node]);
        if (!t.length[f.node].empty()) out += ":" + t.length[f.node];
        if (comment) {
            std::string cm = comment((size_t)f.node);
            if (!cm.empty()) out += "[" + cm + "]";
        }
        st.pop_back();
    }
    out.push_back(';');
    return out;
}

} // namespace qsh
