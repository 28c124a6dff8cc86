#!/usr/bin/env python
"""TEST INFRASTRUCTURE (oracle/seam): generates a header-only stand-in for protoc's convnet_config.pb.h from the reference's
proto/convnet_config.proto, so the reference's operator sources (src/*_edge.cc, layer.cc, optimizer.cc, loss_functions.cc)
compile UNMODIFIED in a container without protoc / libprotobuf.

    python oracle/seam/gen_config_pb.py /root/reference/proto/convnet_config.proto oracle/_ref/gen/convnet_config.pb.h

Only the generated-code surface those sources use is emitted: nested enums (config::Layer::RECTIFIED_LINEAR …), scalar /
string / enum / message / repeated accessors (x(), has_x(), set_x(), x_size(), x(i), add_x(), mutable_x()), CopyFrom and
MergeFrom with proto2 semantics.  No wire format, no reflection, no text parser (the seam driver fills configs in code).
The output is derived from the reference's proto, so it is written under oracle/_ref/ and never committed."""
import re
import sys

SCALARS = {"int32": "int", "int64": "long long", "uint32": "unsigned", "uint64": "unsigned long long", "float": "float",
           "double": "double", "bool": "bool", "string": "std::string"}


LEXER = r"""
// minimal protobuf text-format reader for the generated classes (what ReadPbtxt needs): `key: value`, `key { ... }`,
// `key < ... >`, # comments, quoted strings, numbers, true/false, enum identifiers
namespace seam_pb {
class Lexer {
 public:
  explicit Lexer(const std::string& text) : s_(text), i_(0) {}
  void skip() {
    for (;;) {
      while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r' || s_[i_] == ',' || s_[i_] == ';')) ++i_;
      if (i_ < s_.size() && s_[i_] == '#') { while (i_ < s_.size() && s_[i_] != '\n') ++i_; continue; }
      return;
    }
  }
  bool more(bool nested) {
    skip();
    if (i_ >= s_.size()) { if (nested) fail("unexpected end of input"); return false; }
    if (s_[i_] == '}' || s_[i_] == '>') { if (!nested) fail("unbalanced close"); ++i_; return false; }
    return true;
  }
  std::string ident() {
    skip();
    size_t b = i_;
    while (i_ < s_.size() && (isalnum((unsigned char)s_[i_]) || s_[i_] == '_')) ++i_;
    if (b == i_) fail("identifier expected");
    return s_.substr(b, i_ - b);
  }
  void colon() { skip(); if (i_ < s_.size() && s_[i_] == ':') ++i_; }
  void open() { colon(); skip(); if (i_ < s_.size() && (s_[i_] == '{' || s_[i_] == '<')) ++i_; else fail("'{' expected"); }
  double number() {
    colon(); skip();
    char* end = nullptr;
    const double v = strtod(s_.c_str() + i_, &end);
    if (end == s_.c_str() + i_) fail("number expected");
    i_ = end - s_.c_str();
    if (i_ < s_.size() && (s_[i_] == 'f' || s_[i_] == 'F')) ++i_;
    return v;
  }
  bool boolean() { colon(); const std::string w = ident(); if (w == "true" || w == "1") return true; if (w == "false" || w == "0") return false; fail("bool expected"); return false; }
  std::string word() { colon(); return ident(); }
  std::string str() {
    colon(); skip();
    if (i_ >= s_.size() || (s_[i_] != '"' && s_[i_] != '\'')) fail("string expected");
    const char q = s_[i_++];
    std::string out;
    while (i_ < s_.size() && s_[i_] != q) {
      if (s_[i_] == '\\' && i_ + 1 < s_.size()) { ++i_; out += s_[i_] == 'n' ? '\n' : s_[i_]; } else out += s_[i_];
      ++i_;
    }
    ++i_;
    return out;
  }
  [[noreturn]] void fail(const std::string& what) {
    int line = 1;
    for (size_t k = 0; k < i_ && k < s_.size(); ++k) line += s_[k] == '\n';
    fprintf(stderr, "pbtxt parse error near line %d: %s\n", line, what.c_str());
    exit(1);
  }
 private:
  std::string s_;
  size_t i_;
};
}  // namespace seam_pb
"""


class Msg:
    def __init__(self, name, parent=None):
        self.name, self.parent = name, parent
        self.enums, self.fields, self.nested = [], [], []   # enums: (name, [(val, num)]); fields: dict

    def cpp_name(self):
        return self.name if self.parent is None else self.parent.cpp_name() + "_" + self.name


def parse(text):
    text = re.sub(r"//[^\n]*", "", text)
    toks = re.findall(r'"[^"]*"|[A-Za-z_][\w.]*|-?\d+\.?\d*(?:[eE][-+]?\d+)?|[{}=;\[\],]', text)
    pos = 0
    top = []

    def block(parent):
        nonlocal pos
        name = toks[pos]; pos += 1
        assert toks[pos] == "{"; pos += 1
        m = Msg(name, parent)
        while toks[pos] != "}":
            t = toks[pos]
            if t == "message":
                pos += 1
                m.nested.append(block(m))
            elif t == "enum":
                pos += 1
                ename = toks[pos]; pos += 2
                vals = []
                while toks[pos] != "}":
                    vals.append((toks[pos], int(toks[pos + 2])))
                    pos += 4
                pos += 1
                m.enums.append((ename, vals))
            elif t in ("optional", "required", "repeated"):
                label, ftype, fname = toks[pos], toks[pos + 1], toks[pos + 2]
                pos += 5          # label type name = number
                default = None
                if toks[pos] == "[":
                    while toks[pos] != "]":
                        if toks[pos] == "default":
                            default = toks[pos + 2]
                        pos += 1
                    pos += 1
                assert toks[pos] == ";", (fname, toks[pos])
                pos += 1
                m.fields.append(dict(label=label, type=ftype, name=fname, default=default))
            else:
                raise SystemExit(f"unexpected token {t!r} in message {name}")
        pos += 1
        return m

    while pos < len(toks):
        t = toks[pos]
        if t == "package":
            pos += 3
        elif t == "message":
            pos += 1
            top.append(block(None))
        else:
            raise SystemExit(f"unexpected top-level token {t!r}")
    return top


def flatten(msgs):
    out = []
    for m in msgs:
        out.extend(flatten(m.nested))
        out.append(m)
    return out


def emit(top):
    allm = flatten(top)
    by_name = {}
    for m in allm:
        by_name[m.name] = m
        by_name[m.cpp_name()] = m

    def resolve(ftype, scope):
        """-> (kind, cpp type)"""
        if ftype in SCALARS:
            return ("string" if ftype == "string" else "scalar"), SCALARS[ftype]
        s = scope
        while s is not None:                      # enum / nested message visible from this scope
            for en, _ in s.enums:
                if en == ftype:
                    return "enum", s.cpp_name() + "::" + en
            for n in s.nested:
                if n.name == ftype:
                    return "message", n.cpp_name()
            s = s.parent
        if ftype in by_name:
            return "message", by_name[ftype].cpp_name()
        raise SystemExit(f"unknown type {ftype}")

    # order: dependencies first
    order, seen = [], set()

    def visit(m):
        if m.cpp_name() in seen:
            return
        seen.add(m.cpp_name())
        for f in m.fields:
            kind, cpp = resolve(f["type"], m)
            if kind == "message":
                visit(by_name[cpp])
        order.append(m)
    for m in allm:
        visit(m)

    o = ["// GENERATED by oracle/seam/gen_config_pb.py from the reference's proto/convnet_config.proto — test infrastructure, not committed.",
         "#pragma once", "#include <cstdio>", "#include <cstdlib>", "#include <string>", "#include <vector>", "", LEXER, "namespace config {", ""]
    for m in order:
        cn = m.cpp_name()
        o.append(f"class {cn} {{")
        o.append(" public:")
        for en, vals in m.enums:
            o.append(f"  enum {en} {{ " + ", ".join(f"{v} = {n}" for v, n in vals) + " };")
            o.append(f"  static bool Parse_{cn}_{en}(const std::string& w, {en}* out) {{")
            for v, n in vals:
                o.append(f'    if (w == "{v}") {{ *out = {v}; return true; }}')
            o.append("    return false;")
            o.append("  }")
        for n in m.nested:
            o.append(f"  typedef {n.cpp_name()} {n.name};")
        members, copy, merge, parse = [], [], [], []
        for f in m.fields:
            kind, cpp = resolve(f["type"], m)
            n, rep = f["name"], f["label"] == "repeated"
            if rep:
                members.append(f"  std::vector<{cpp}> {n}_;")
                o.append(f"  int {n}_size() const {{ return (int){n}_.size(); }}")
                o.append(f"  const {cpp}& {n}(int i) const {{ return {n}_[i]; }}")
                o.append(f"  const std::vector<{cpp}>& {n}() const {{ return {n}_; }}")
                o.append(f"  std::vector<{cpp}>* mutable_{n}() {{ return &{n}_; }}")
                o.append(f"  void clear_{n}() {{ {n}_.clear(); }}")
                if kind == "message":
                    o.append(f"  {cpp}* add_{n}() {{ {n}_.emplace_back(); return &{n}_.back(); }}")
                    o.append(f"  {cpp}* mutable_{n}(int i) {{ return &{n}_[i]; }}")
                else:
                    o.append(f"  void add_{n}(const {cpp}& v) {{ {n}_.push_back(v); }}")
                    o.append(f"  void set_{n}(int i, const {cpp}& v) {{ {n}_[i] = v; }}")
                merge.append(f"    {n}_.insert({n}_.end(), o.{n}_.begin(), o.{n}_.end());")
                if kind == "message":
                    parse.append(f'    if (key == "{n}") {{ lx.open(); add_{n}()->ParseText(lx, true); return true; }}')
                elif kind == "string":
                    parse.append(f'    if (key == "{n}") {{ add_{n}(lx.str()); return true; }}')
                elif kind == "enum":
                    parse.append(f'    if (key == "{n}") {{ {cpp} v; if (!Parse_{cpp.replace("::", "_")}(lx.word(), &v)) lx.fail("bad enum value for {n}"); add_{n}(v); return true; }}')
                elif cpp == "bool":
                    parse.append(f'    if (key == "{n}") {{ add_{n}(lx.boolean()); return true; }}')
                else:
                    parse.append(f'    if (key == "{n}") {{ add_{n}(({cpp})lx.number()); return true; }}')
                continue
            d = f["default"]
            if kind == "scalar":
                init = {"bool": "false"}.get(cpp, "0") if d is None else d
                if cpp == "float" and d is not None and re.fullmatch(r"-?\d+", d):
                    init = d + ".0f"
                elif cpp == "float" and d is not None:
                    init = d + "f"
            elif kind == "string":
                init = '""' if d is None else d
            elif kind == "enum":
                first = None
                s = m
                en = f["type"]
                while s is not None and first is None:
                    for ename, vals in s.enums:
                        if ename == en:
                            first = vals[0][0]
                    s = s.parent
                init = f"{cpp.rsplit('::', 1)[0]}::{d if d is not None else first}"
            if kind == "message":
                members.append(f"  {cpp} {n}_; bool has_{n}_ = false;")
                o.append(f"  const {cpp}& {n}() const {{ return {n}_; }}")
                o.append(f"  {cpp}* mutable_{n}() {{ has_{n}_ = true; return &{n}_; }}")
                o.append(f"  bool has_{n}() const {{ return has_{n}_; }}")
                o.append(f"  void clear_{n}() {{ {n}_ = {cpp}(); has_{n}_ = false; }}")
                merge.append(f"    if (o.has_{n}_) {{ has_{n}_ = true; {n}_.MergeFrom(o.{n}_); }}")
                parse.append(f'    if (key == "{n}") {{ lx.open(); mutable_{n}()->ParseText(lx, true); return true; }}')
            else:
                argt = f"const {cpp}&" if kind == "string" else cpp
                members.append(f"  {cpp} {n}_ = {init}; bool has_{n}_ = false;")
                o.append(f"  {argt} {n}() const {{ return {n}_; }}")
                o.append(f"  void set_{n}({argt} v) {{ {n}_ = v; has_{n}_ = true; }}")
                o.append(f"  bool has_{n}() const {{ return has_{n}_; }}")
                o.append(f"  void clear_{n}() {{ {n}_ = {init}; has_{n}_ = false; }}")
                if kind == "string":
                    o.append(f"  std::string* mutable_{n}() {{ has_{n}_ = true; return &{n}_; }}")
                merge.append(f"    if (o.has_{n}_) {{ {n}_ = o.{n}_; has_{n}_ = true; }}")
                if kind == "string":
                    parse.append(f'    if (key == "{n}") {{ set_{n}(lx.str()); return true; }}')
                elif kind == "enum":
                    parse.append(f'    if (key == "{n}") {{ {cpp} v; if (!Parse_{cpp.replace("::", "_")}(lx.word(), &v)) lx.fail("bad enum value for {n}"); set_{n}(v); return true; }}')
                elif cpp == "bool":
                    parse.append(f'    if (key == "{n}") {{ set_{n}(lx.boolean()); return true; }}')
                else:
                    parse.append(f'    if (key == "{n}") {{ set_{n}(({cpp})lx.number()); return true; }}')
        o.append(f"  void CopyFrom(const {cn}& o) {{ *this = o; }}")
        o.append(f"  void MergeFrom(const {cn}& o) {{")
        o.extend(merge)
        o.append("  }")
        o.append(f"  void Clear() {{ *this = {cn}(); }}")
        o.append("  bool ParseTextField(const std::string& key, ::seam_pb::Lexer& lx) {")
        o.extend(parse)
        o.append("    return false;")
        o.append("  }")
        o.append("  void ParseText(::seam_pb::Lexer& lx, bool nested) {")
        o.append('    while (lx.more(nested)) { const std::string key = lx.ident(); if (!ParseTextField(key, lx)) lx.fail("unknown field " + key); }')
        o.append("  }")
        o.append("  void ParseFromText(const std::string& text) { ::seam_pb::Lexer lx(text); ParseText(lx, false); }")
        o.append('  std::string DebugString() const { return "<config::' + cn + '>"; }')
        o.append(" private:")
        o.extend(members)
        o.append("};")
        o.append("")
    o.append("}  // namespace config")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    import os
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write(emit(parse(open(src).read())))
    print("wrote", dst)
