// Regex-circuit loader: a circom-2 front end for templates of the kind zk-regex generates.
//
// EmailVerifier instantiates `BodyHashRegex(maxHeadersLength)` from the npm package
// @zk-email/zk-regex-circom (packages/circuits/email-verifier.circom:5,126-127); that generated file is not
// part of the reference tree.  Instead of hard-wiring one circuit, the schedule can be built FROM the
// template text: this header parses the supplied `.circom` file (and what it includes), elaborates the
// template for the concrete `msg_bytes`, and lowers every signal that carries information -- hints (`<--`)
// and signals assigned a quadratic expression, the kept-v1 rule of zkwg_layout.h -- into a gate list over
// small integers.  zk_net_eval (zkwg_kernels_net.hip) evaluates that list per email, zk_expand's ZSEG_NET
// streams the values out.  Linear signals are substituted away (they are what `--O1/--O2` removes; an O0 build
// gets them back from the `.r1cs`, zkwg_full.h).
//
// Language subset: templates with integer parameters, integer functions, `signal` / `component` / `var` declarations (arrays,
// initialisers), `for` / `while` / `if`, `<==` `==>` `<--` `===`, component arrays, anonymous components
// `T(p)(in...)` named `<T>_<line>_<offset>` like the compiler's syntax-sugar remover, array literals, integer
// `var` arithmetic.  Hints must have one of circomlib's two shapes: `(x >> k) & 1` (Num2Bits) and
// `x != 0 ? 1/x : 0` (IsZero).  Values are small signed integers (|v| < 2^30, checked by interval analysis at
// load time) or the inverse of one; anything else is refused with file:line.
// circomlib's comparators / gates / bitify are read from the include path when present and otherwise
// taken from the restatement at the end of this file ([EXT] circomlib 2.0.5).
#pragma once
#include <chrono>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace zkc {

typedef long long i64;
typedef uint32_t u32;
typedef uint8_t u8;

struct Error { std::string msg; };
[[noreturn]] static inline void fail(const std::string& m) { throw Error{m}; }

// ------------------------------------------------------------------------------------------------ lexer
enum { T_ID, T_NUM, T_STR, T_OP, T_EOF };
struct Tok { int kind; std::string s; int line; int pos; };

static inline void lex(const std::string& text, const std::string& fname, std::vector<Tok>& out) {
  static const char* ops3[] = {"<==", "==>", "<--", "-->", "===", "**=", "<<=", ">>="};
  static const char* ops2[] = {"++", "--", "+=", "-=", "*=", "/=", "\\=", "%=", "&=", "|=", "^=", "==", "!=",
                               "<=", ">=", "&&", "||", "<<", ">>", "**"};
  size_t i = 0, n = text.size();
  int line = 1;
  while (i < n) {
    char c = text[i];
    if (c == '\n') { ++line; ++i; continue; }
    if (c == ' ' || c == '\t' || c == '\r') { ++i; continue; }
    if (c == '/' && i + 1 < n && text[i + 1] == '/') { while (i < n && text[i] != '\n') ++i; continue; }
    if (c == '/' && i + 1 < n && text[i + 1] == '*') {
      i += 2;
      while (i + 1 < n && !(text[i] == '*' && text[i + 1] == '/')) { if (text[i] == '\n') ++line; ++i; }
      i += 2;
      continue;
    }
    if (isalpha((unsigned char)c) || c == '_' || c == '$') {
      size_t j = i;
      while (j < n && (isalnum((unsigned char)text[j]) || text[j] == '_' || text[j] == '$')) ++j;
      out.push_back(Tok{T_ID, text.substr(i, j - i), line, (int)i});
      i = j;
      continue;
    }
    if (isdigit((unsigned char)c)) {
      size_t j = i;
      if (c == '0' && j + 1 < n && (text[j + 1] == 'x' || text[j + 1] == 'X')) { j += 2; while (j < n && isxdigit((unsigned char)text[j])) ++j; }
      else while (j < n && isdigit((unsigned char)text[j])) ++j;
      out.push_back(Tok{T_NUM, text.substr(i, j - i), line, (int)i});
      i = j;
      continue;
    }
    if (c == '"') {
      size_t j = i + 1;
      while (j < n && text[j] != '"') ++j;
      out.push_back(Tok{T_STR, text.substr(i + 1, j - i - 1), line, (int)i});
      i = j + 1;
      continue;
    }
    bool done = false;
    for (const char* o : ops3) if (text.compare(i, 3, o) == 0) { out.push_back(Tok{T_OP, o, line, (int)i}); i += 3; done = true; break; }
    if (done) continue;
    for (const char* o : ops2) if (text.compare(i, 2, o) == 0) { out.push_back(Tok{T_OP, o, line, (int)i}); i += 2; done = true; break; }
    if (done) continue;
    if (strchr("+-*/\\%&|^~!<>=?:;,.()[]{}", c)) { out.push_back(Tok{T_OP, std::string(1, c), line, (int)i}); ++i; continue; }
    fail(fname + ":" + std::to_string(line) + ": unexpected character '" + std::string(1, c) + "'");
  }
  out.push_back(Tok{T_EOF, "", line, (int)n});
}

// ------------------------------------------------------------------------------------------------ AST
enum NodeKind {
  N_NUM, N_ID, N_INDEX, N_MEMBER, N_CALL, N_ANON, N_ARR, N_BIN, N_UN, N_TERN,
  S_BLOCK, S_SIGNAL, S_COMP, S_VAR, S_FOR, S_WHILE, S_IF, S_RET, S_ASSERT, S_LOG, S_ASSIGN, S_CONSTR, S_NOP,
  N_DECL, N_LIST
};
struct Node {
  int k = S_NOP;
  std::string s;          // identifier / operator / signal kind
  i64 n = 0;              // N_NUM value
  bool big = false;       // N_NUM does not fit 62 bits (only legal where it is never evaluated)
  std::vector<Node*> c;   // children
  int line = 0, pos = 0;
  const std::string* file = nullptr;
};

struct Template { std::string name; std::vector<std::string> params; Node* body = nullptr; const std::string* file = nullptr; };

struct Parser {
  std::vector<Tok> t;
  size_t i = 0;
  const std::string* file;
  std::deque<Node>& pool;
  Parser(std::deque<Node>& p, const std::string* f) : file(f), pool(p) {}

  Node* mk(int k, const Tok& at) { pool.emplace_back(); Node* n = &pool.back(); n->k = k; n->line = at.line; n->pos = at.pos; n->file = file; return n; }
  const Tok& cur() const { return t[i]; }
  bool at(const char* s) const { return (t[i].kind == T_OP || t[i].kind == T_ID) && t[i].s == s; }
  bool accept(const char* s) { if (at(s)) { ++i; return true; } return false; }
  [[noreturn]] void err(const std::string& m) const { fail(*file + ":" + std::to_string(t[i].line) + ": " + m + " (at '" + t[i].s + "')"); }
  void expect(const char* s) { if (!accept(s)) err(std::string("expected '") + s + "'"); }
  std::string ident() { if (t[i].kind != T_ID) err("expected an identifier"); return t[i++].s; }

  // ---- expressions (precedence climbing)
  Node* expression() { return ternary(); }
  Node* ternary() {
    Node* c = binary(0);
    if (at("?")) {
      Node* n = mk(N_TERN, cur());
      ++i;
      Node* a = ternary();
      expect(":");
      Node* b = ternary();
      n->c = {c, a, b};
      return n;
    }
    return c;
  }
  static int level_of(const std::string& op) {
    static const std::vector<std::vector<const char*>> L = {
        {"||"}, {"&&"}, {"==", "!=", "<", ">", "<=", ">="}, {"|"}, {"^"}, {"&"}, {"<<", ">>"}, {"+", "-"}, {"*", "/", "\\", "%"}, {"**"}};
    for (size_t l = 0; l < L.size(); ++l) for (const char* o : L[l]) if (op == o) return (int)l;
    return -1;
  }
  Node* binary(int lvl) {
    if (lvl > 9) return unary();
    Node* l = binary(lvl + 1);
    while (t[i].kind == T_OP && level_of(t[i].s) == lvl) {
      Node* n = mk(N_BIN, cur());
      n->s = t[i++].s;
      Node* r = binary(lvl + 1);
      n->c = {l, r};
      l = n;
    }
    return l;
  }
  Node* unary() {
    if (t[i].kind == T_OP && (t[i].s == "-" || t[i].s == "!" || t[i].s == "~")) {
      Node* n = mk(N_UN, cur());
      n->s = t[i++].s;
      n->c = {unary()};
      return n;
    }
    return postfix();
  }
  Node* list_until(const char* close) {
    Node* l = mk(N_LIST, cur());
    while (!at(close)) {
      l->c.push_back(expression());
      if (!accept(",")) break;
    }
    expect(close);
    return l;
  }
  Node* postfix() {
    Node* b = nullptr;
    if (t[i].kind == T_NUM) {
      b = mk(N_NUM, cur());
      const std::string& s = t[i].s;
      unsigned long long v = 0;
      bool big = false;
      if (s.size() > 2 && (s[1] == 'x' || s[1] == 'X')) { if (s.size() > 17) big = true; else v = strtoull(s.c_str() + 2, nullptr, 16); }
      else { if (s.size() > 18) big = true; else v = strtoull(s.c_str(), nullptr, 10); }
      if (v >> 62) big = true;
      b->n = (i64)v; b->big = big; b->s = s;
      ++i;
    } else if (at("(")) {
      ++i;
      b = expression();
      expect(")");
    } else if (at("[")) {
      Node* n = mk(N_ARR, cur());
      ++i;
      Node* l = list_until("]");
      n->c = l->c;
      b = n;
    } else if (t[i].kind == T_ID) {
      Tok id = t[i++];
      if (at("(")) {
        ++i;
        Node* args = list_until(")");
        if (at("(")) {           // anonymous component T(params)(inputs)
          ++i;
          Node* ins = list_until(")");
          Node* n = mk(N_ANON, id);
          n->s = id.s;
          n->c = {args, ins};
          b = n;
        } else {
          Node* n = mk(N_CALL, id);
          n->s = id.s;
          n->c = args->c;
          b = n;
        }
      } else {
        b = mk(N_ID, id);
        b->s = id.s;
      }
    } else err("expected an expression");
    for (;;) {
      if (at("[")) {
        Node* n = mk(N_INDEX, cur());
        ++i;
        Node* e = expression();
        expect("]");
        n->c = {b, e};
        b = n;
      } else if (at(".")) {
        Node* n = mk(N_MEMBER, cur());
        ++i;
        n->s = ident();
        n->c = {b};
        b = n;
      } else break;
    }
    return b;
  }

  // ---- statements
  Node* block() {
    Node* n = mk(S_BLOCK, cur());
    expect("{");
    while (!at("}")) n->c.push_back(statement());
    expect("}");
    return n;
  }
  Node* decls(int kind, const std::string& skind) {
    Node* n = mk(kind, cur());
    n->s = skind;
    for (;;) {
      Node* d = mk(N_DECL, cur());
      d->s = ident();
      Node* dims = mk(N_LIST, cur());
      while (at("[")) { ++i; dims->c.push_back(expression()); expect("]"); }
      Node* init = nullptr;
      std::string op;
      if (at("=") || at("<==") || at("<--")) { op = t[i++].s; init = expression(); }
      d->c = {dims};
      if (init) { d->c.push_back(init); Node* o = mk(N_ID, cur()); o->s = op; d->c.push_back(o); }
      n->c.push_back(d);
      if (!accept(",")) break;
    }
    return n;
  }
  Node* simple() {   // assignment / constraint / increment (no trailing ';')
    Tok at0 = cur();
    Node* l = expression();
    if (t[i].kind == T_OP) {
      const std::string op = t[i].s;
      if (op == "++" || op == "--") {
        ++i;
        Node* one = mk(N_NUM, at0); one->n = 1;
        Node* n = mk(S_ASSIGN, at0); n->s = op == "++" ? "+=" : "-="; n->c = {l, one};
        return n;
      }
      if (op == "===") { ++i; Node* r = expression(); Node* n = mk(S_CONSTR, at0); n->c = {l, r}; return n; }
      if (op == "==>" || op == "-->") {
        ++i;
        Node* r = expression();
        Node* n = mk(S_ASSIGN, at0); n->s = op == "==>" ? "<==" : "<--"; n->c = {r, l};
        return n;
      }
      static const char* as[] = {"=", "<==", "<--", "+=", "-=", "*=", "/=", "\\=", "%=", "**=", "<<=", ">>=", "&=", "|=", "^="};
      for (const char* a : as) if (op == a) {
        ++i;
        Node* r = expression();
        Node* n = mk(S_ASSIGN, at0); n->s = op; n->c = {l, r};
        return n;
      }
    }
    Node* n = mk(S_NOP, at0);   // expression statement without effect
    return n;
  }
  Node* statement() {
    if (at("{")) return block();
    if (at("signal")) {
      ++i;
      std::string kind = "mid";
      if (accept("input")) kind = "in"; else if (accept("output")) kind = "out";
      if (at("{")) { while (!at("}")) ++i; ++i; }   // tags
      Node* n = decls(S_SIGNAL, kind);
      expect(";");
      return n;
    }
    if (at("component")) { ++i; Node* n = decls(S_COMP, ""); expect(";"); return n; }
    if (at("var")) { ++i; Node* n = decls(S_VAR, ""); expect(";"); return n; }
    if (at("for")) {
      Node* n = mk(S_FOR, cur());
      ++i;
      expect("(");
      Node* init;
      if (at("var")) { ++i; init = decls(S_VAR, ""); } else init = simple();
      expect(";");
      Node* cond = expression();
      expect(";");
      Node* step = simple();
      expect(")");
      Node* body = statement();
      n->c = {init, cond, step, body};
      return n;
    }
    if (at("while")) {
      Node* n = mk(S_WHILE, cur());
      ++i;
      expect("(");
      Node* cond = expression();
      expect(")");
      n->c = {cond, statement()};
      return n;
    }
    if (at("if")) {
      Node* n = mk(S_IF, cur());
      ++i;
      expect("(");
      Node* cond = expression();
      expect(")");
      Node* a = statement();
      n->c = {cond, a};
      if (accept("else")) n->c.push_back(statement());
      return n;
    }
    if (at("return")) { Node* n = mk(S_RET, cur()); ++i; n->c = {expression()}; expect(";"); return n; }
    if (at("assert")) { Node* n = mk(S_ASSERT, cur()); ++i; expect("("); n->c = {expression()}; expect(")"); expect(";"); return n; }
    if (at("log")) {
      Node* n = mk(S_LOG, cur());
      ++i;
      expect("(");
      int depth = 1;
      while (depth > 0 && t[i].kind != T_EOF) { if (at("(")) ++depth; else if (at(")")) --depth; ++i; }
      expect(";");
      return n;
    }
    Node* n = simple();
    expect(";");
    return n;
  }
};

// ------------------------------------------------------------------------------------------------ values
struct Lin {
  i64 c0 = 0;
  std::vector<std::pair<u32, i64>> t;   // (source, coefficient), sorted by source
  bool sig = false;                     // derived from a signal: degree 1 for the compiler even when its value is a constant
  bool is_const() const { return t.empty() && !sig; }
};
static const u32 SRC_INPUT = 0x80000000u;   // source = message byte (index in the low bits)
static const i64 COEF_LIMIT = (i64)1 << 40;

static inline i64 chk(__int128 v, const char* what) {
  if (v >= (__int128)COEF_LIMIT || v <= -(__int128)COEF_LIMIT) fail(std::string("constant out of range in ") + what);
  return (i64)v;
}
static inline Lin lin_const(i64 c) { Lin l; l.c0 = c; return l; }
static inline Lin lin_add(const Lin& x, const Lin& y, i64 sy = 1) {
  Lin r;
  r.sig = x.sig || y.sig;
  r.c0 = chk((__int128)x.c0 + (__int128)sy * y.c0, "a sum");
  r.t.reserve(x.t.size() + y.t.size());
  size_t i = 0, j = 0;
  while (i < x.t.size() || j < y.t.size()) {
    if (j == y.t.size() || (i < x.t.size() && x.t[i].first < y.t[j].first)) r.t.push_back(x.t[i++]);
    else if (i == x.t.size() || y.t[j].first < x.t[i].first) { r.t.emplace_back(y.t[j].first, chk((__int128)sy * y.t[j].second, "a sum")); ++j; }
    else {
      i64 c = chk((__int128)x.t[i].second + (__int128)sy * y.t[j].second, "a sum");
      if (c) r.t.emplace_back(x.t[i].first, c);
      ++i; ++j;
    }
  }
  return r;
}
static inline Lin lin_scale(const Lin& x, i64 k) {
  Lin r;
  if (k == 0) return r;
  r.sig = x.sig;
  r.c0 = chk((__int128)x.c0 * k, "a product");
  for (auto& p : x.t) r.t.emplace_back(p.first, chk((__int128)p.second * k, "a product"));
  return r;
}
static inline bool lin_eq(const Lin& x, const Lin& y) { return x.c0 == y.c0 && x.t == y.t; }
// x == k * y for some integer k?
static inline bool lin_multiple(const Lin& x, const Lin& y, i64& k) {
  if (y.t.empty()) {
    if (!x.t.empty()) return false;
    if (y.c0 == 0) { k = 0; return x.c0 == 0; }
    if (x.c0 % y.c0) return false;
    k = x.c0 / y.c0;
    return true;
  }
  if (x.t.size() != y.t.size()) return false;
  if (x.t[0].second % y.t[0].second) return false;
  k = x.t[0].second / y.t[0].second;
  return lin_eq(x, lin_scale(y, k));
}

enum { V_UNSET, V_LIN, V_QUAD, V_ARR };
struct Val {
  int k = V_UNSET;
  Lin a, b, c;              // V_LIN: a    V_QUAD: a * b + c
  std::vector<Val> arr;     // V_ARR
  static Val lin(const Lin& l) { Val v; v.k = V_LIN; v.a = l; return v; }
  static Val num(i64 n) { return lin(lin_const(n)); }
  bool is_const() const { return k == V_LIN && a.is_const(); }
};

// ------------------------------------------------------------------------------------------------ gates
enum GateOp : u32 {
  G_NOP = 0,
  G_QUAD = 1,    // v = A * B + C
  G_INV0 = 2,    // v = A == 0 ? 0 : 1 / A           (stored as the integer A with the inverse flag)
  G_BIT = 3,     // v = (A >> k) & 1
  G_NEZ = 4,     // v = k * (A != 0) + C             (IsZero.out = -in * inv + 1 and relatives)
  G_LIN = 5,     // v = A                            (materialised long linear form; not a signal of the layout)
  G_ASSERT = 6,  // A * B + C == 0
  G_OUT = 7      // small[dst] = A                   (template outputs handed to the rest of the schedule)
};
struct Inst;
struct Gate {
  u32 op = G_NOP;
  i64 k = 0;
  Lin f[3];
  Inst* owner = nullptr;   // kept signal: the component, signal and flat index it belongs to
  u32 sig = 0, flat = 0;
  i64 lo = 0, hi = 0;      // value interval
  u32 slot = 0xffffffffu;  // layout slot inside the region (kept gates first, then temporaries)
  u32 out_index = 0;       // G_OUT: index in the output list
  const Node* at = nullptr;
};

struct SigArr {
  std::string name;
  int kind = 0;                 // 0 out, 1 in, 2 mid
  std::vector<u32> dims;
  std::vector<Lin> val;
  std::vector<u8> set;
  std::vector<int> gate;        // gate id when the signal itself is a gate (kept), else -1
  u32 size() const { u32 n = 1; for (u32 d : dims) n *= d; return n; }
};
struct CompArr { std::vector<u32> dims; std::vector<Inst*> inst; };
struct VarArr { std::vector<u32> dims; std::vector<Val> v; };

struct Inst {
  const Template* tmpl = nullptr;
  std::vector<i64> args;
  std::string name;             // name inside the parent ("eq[0][5]", "MultiOR_351_24[3]")
  Inst* parent = nullptr;
  std::vector<Inst*> subs;      // creation order
  std::vector<SigArr> sigs;     // declaration order
  std::map<std::string, u32> sig_of;
  std::map<std::string, CompArr> comps;
  u32 pending = 0;              // input elements not assigned yet
  bool ran = false, running = false;
};

struct Frame {
  Inst* inst;
  std::map<std::string, VarArr> vars;
  std::vector<int> loops;       // iteration counters of the enclosing loops (innermost last)
};

// ------------------------------------------------------------------------------------------------ result
struct ChainTab {
  u32 end = 0;                       // positions covered, in walking order: bytes [0, end) forward, [n_in - end, n_in) backward (0: no chain)
  u32 smax = 0;                      // states: rows of every table
  u32 classes = 0;                   // distinct block descriptors; a periodic circuit has a handful
  u32 mask_words = 0;                // mask words per position served from `mask`
  u32 fdim = 1;                      // symbols per position = fdim x 256 (backward pass: fdim = the forward chain's states)
  std::vector<u8> cls;               // [n_in] class of a message byte (bytes outside the chain: 0, unused)
  std::vector<u8> delta;             // [classes][smax][fdim][256] next state
  std::vector<u32> mask;             // [classes][smax][fdim][256][mask_words]
  std::vector<u32> tab;              // [tables][smax][fdim][256] stored words
  std::vector<u32> reach_bits;       // [n_in][8] host only: the states that can enter a byte (the backward pass walks consistent symbols only)
  u32 n_gates = 0, n_front = 0;      // statistics: gates served from the tables / of those, read by the list through mask bits
};
struct Net {
  u32 n_in = 0;                      // message bytes
  u32 n_kept = 0, n_temp = 0;        // region = kept slots, then temporaries
  u32 inv_need = 0;                  // largest |x| whose inverse a kept signal may hold (interval bound, capped at 2^16;
                                     // the evaluator rejects an email whose value exceeds the table)
  u32 n_out = 0;                     // outputs: [0] = first scalar output, then the elements of the first array output
  u32 lds_log2 = 0;                  // (unused)
  u32 n_pins = 0;                    // LDS words holding gate values (allocated by liveness)
  u32 lds_words = 0;                 // LDS image of the evaluator: values, message bytes, a zero word, a scratch word
  u32 n_general = 0;                 // gates on the evaluator's 64-bit path (the 32-bit path is not provably exact for them)
  std::vector<u32> records;          // 16 words per gate, in execution order (zkwg_net_core.h)
  std::vector<u32> step_count;       // gates per step (<= 64, one per lane) | 0x8000 general path | 0x4000 term slots 0..3 only;
                                     // padded with empty steps (the evaluator reads the counts three groups of 8 ahead)
  u32 n_steps = 0;
  // Byte-local signals (zkwg_circom.h localize): a signal whose value depends on ONE message byte only (the character
  // comparators of a regex circuit: IsEqual / LessThan internals, range ANDs, class ORs -- most of its signals) is a
  // function of that byte.  Such gates are not gates of the list: zk_net_fill writes their words from `fn_tab` (the stored word for
  // each of the 256 byte values, one table per distinct function), and the boolean ones a per-email gate does read come from
  // a per-position mask word the evaluator builds with one lookup of `mask_tab` per message byte.
  std::vector<u32> fn_tab;           // n_fn x 256 stored words
  std::vector<u32> slot_desc;        // per kept slot: 0 = evaluated (the word is in the image), 0x80000000 | fn << 16 | byte index
                                     // (byte-local), 0xC0000000 | table << 16 | byte index (chain.tab) or 0xE0000000 | ... (bchain.tab)
  std::vector<u32> mask_tab;         // 256 x mask_words: bit b of word m = truth of frontier function 23 m + b on that byte value
  u32 mask_words = 0;                // mask words per message byte (0: nothing was localised)
  u32 lds_masks = 0;                 // first LDS word of the evaluator's mask region (n_in x mask_words words)
  u32 n_local = 0, n_frontier = 0;   // statistics: gates removed from the evaluator / served from the masks
  // Recurrences collapsed to scans (zkwg_circom.h chain_pass): the gates that depend on the bytes 0 .. i through a bounded set of
  // carried values (the state vector of a regex circuit: AND -> MultiOR -> states[i + 1]) are functions of (state entering byte i,
  // byte i), the state being the index of the carried valuation among the reachable ones.  zk_net_scan walks
  // state' = delta[class of i][state][byte] per email; zk_net_fill writes the kept ones' words from tab; the boolean ones a gate
  // of the list still reads are bits of mask[class][state][byte].  `bchain`: what then still runs backwards over the message
  // (is_consecutive / live chains), the same with symbols (forward state, byte).
  ChainTab chain, bchain;
  // The region as zk_expand reads it (finish_region): a sequence of periodic RUNS.  Slot start + i * period + q of a run is described by
  // pd entry pd0 + q, taken at position (its own) + i: the circuit instantiates the same components for every message byte, so one
  // period of descriptors serves the whole run (the stand-in at N = 1,024: 96 % of the region is one run of period 220); what is not
  // periodic is a run of one period.  A descriptor names a column of one of three TRANSPOSED tables -- all functions of one row side by
  // side, so that the lanes of a wavefront (neighbouring slots of one position) read neighbouring words:
  //     byte-local  tabs[(byte) * nL + col]          forward  tabs[offF + (fstate * 256 + byte) * nF + col]
  //     backward    tabs[offB + ((bstate * fdim + fstate) * 256 + byte) * nB + col]
  // or says that the evaluator left the word in the image.  zk_expand decodes a table-served slot from the position word
  // (byte | fstate << 8 | bstate << 16, written by zk_net_eval's prologue): the image holds evaluated words only.
  struct Run { u32 start, nslots, period, pd0, pos0, dense; };   // pos0: the position the run's descriptors are relative to
  // dense != 0xffffffff: a long run all of whose slots are byte-local or forward-chain signals of the run's own position has a DENSE table
  // tabs[dense + ((fstate * 256 + byte) * period + q)] -- one row per (forward state, byte), one column per slot of the period -- so that
  // zk_expand reads position word -> table word with no descriptor in between, the lanes of a wavefront reading consecutive words
  std::vector<Run> runs;
  std::vector<u32> pd;               // 2 words per descriptor: [type << 30 | column] [position of period 0 minus the run's pos0 (signed; 0 for nearly all)]; type 0 evaluated, 1 byte-local, 2 forward, 3 backward
  std::vector<u32> tabs;             // transposed tables, L | F | B
  u32 nL = 0, nF = 0, nB = 0, offF = 0, offB = 0;
  u32 lanes = 64;                    // lanes per email of zk_net_eval = gates per step (64 / lanes emails share a wavefront)
  std::vector<std::string> names;    // kept slot -> name relative to the component (".eq[0][5].isz.inv")
  std::vector<u8> boolean;           // kept slot -> 1 if its interval is [0, 1]
  // statistics
  u32 n_gates = 0, n_asserts = 0, n_chunks = 0;
  u64 lds_hits = 0, pin_reads = 0;   // operand reads
};
// record / term encoding shared with the kernel
static const u32 ZKC_MASK_BITS = 23;         // frontier bits per mask word (the evaluator's 32-bit path multiplies signed 24-bit operands)
static const u32 VAL_INVERSE = 0x80000000u; // stored word: inverse of the 31-bit two's-complement integer in the low bits

// ------------------------------------------------------------------------------------------------ elaboration
struct ReturnValue { Val v; };
struct Elab {
  int chain_limit = -1;     // see load()
  int call_depth = 0;
  std::deque<Node> pool;
  std::deque<std::string> files;
  std::deque<Inst> insts;
  std::map<std::string, Template> templates;
  std::map<std::string, Template> functions;   // integer functions (evaluated on compile-time constants)
  std::set<std::string> included;
  std::vector<std::string> include_dirs;
  std::vector<Gate> gates;
  std::vector<u8> g_skip;            // localize(): the gate is not evaluated per email
  std::vector<int> g_front;          // localize(): >= 0: the gate is the bit of that index of its byte's mask words
  std::vector<long long> g_sup;      // localize(): the message byte a gate depends on (-1 none, -2 several)
  std::vector<u8> g_chain;           // chain_pass(): 1 / 2: the gate is a function of (forward / backward chain state, symbol) at its position
  std::vector<int> g_fpos;           // frontier gates: the position whose mask words hold their bit
  std::vector<i64> in_lo, in_hi;
  u32 n_in = 0;
  u32 max_terms = 32;

  // ---- sources
  static const char* builtin_source(const std::string& base);
  bool read_file(const std::string& path, std::string& text) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    text = ss.str();
    return true;
  }
  static std::string dir_of(const std::string& p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? std::string(".") : p.substr(0, k); }
  static std::string base_of(const std::string& p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
  void load_text(const std::string& text, const std::string& fname, const std::string& dir) {
    files.push_back(fname);
    const std::string* fp = &files.back();
    Parser P(pool, fp);
    lex(text, fname, P.t);
    while (P.cur().kind != T_EOF) {
      if (P.accept("pragma")) { while (!P.at(";")) ++P.i; ++P.i; continue; }
      if (P.accept("include")) {
        if (P.cur().kind != T_STR) P.err("expected a file name");
        std::string inc = P.t[P.i++].s;
        P.expect(";");
        load_include(inc, dir);
        continue;
      }
      if (P.at("template")) {
        ++P.i;
        while (P.at("custom") || P.at("parallel")) ++P.i;
        Template T;
        T.name = P.ident();
        T.file = fp;
        P.expect("(");
        while (!P.at(")")) { T.params.push_back(P.ident()); if (!P.accept(",")) break; }
        P.expect(")");
        T.body = P.block();
        templates[T.name] = T;
        continue;
      }
      if (P.at("function")) {
        ++P.i;
        Template F;
        F.name = P.ident();
        F.file = fp;
        P.expect("(");
        while (!P.at(")")) { F.params.push_back(P.ident()); if (!P.accept(",")) break; }
        P.expect(")");
        F.body = P.block();
        functions[F.name] = F;
        continue;
      }
      if (P.at("component")) {   // `component main ... = T(...);`
        while (!P.at(";")) ++P.i;
        ++P.i;
        continue;
      }
      P.err("expected a template, function or include");
    }
  }
  void load_include(const std::string& inc, const std::string& from_dir) {
    std::vector<std::string> cands;
    cands.push_back(from_dir + "/" + inc);
    for (auto& d : include_dirs) cands.push_back(d + "/" + inc);
    for (auto& p : cands) {
      std::string text;
      if (read_file(p, text)) {
        if (!included.insert(base_of(p) + "#" + std::to_string(text.size())).second) return;
        load_text(text, p, dir_of(p));
        return;
      }
    }
    // circomlib: the restatement carried by this library
    const std::string base = base_of(inc);
    if (const char* src = builtin_source(base)) {
      if (!included.insert("builtin:" + base).second) return;
      load_text(src, "<zkwg circomlib>/" + base, "<zkwg circomlib>");
      return;
    }
    fail("include \"" + inc + "\" not found (searched " + from_dir + " and the include directories)");
  }

  // ---- helpers
  std::string where(const Node* n) const { return (n && n->file ? *n->file : std::string("?")) + ":" + std::to_string(n ? n->line : 0); }
  [[noreturn]] void err(const Node* n, const std::string& m) const { fail(where(n) + ": " + m); }

  i64 const_of(const Val& v, const Node* n, const char* what) const {
    if (!v.is_const()) err(n, std::string(what) + " must be a compile-time constant");
    return v.a.c0;
  }

  // ---- arithmetic on values
  Val v_add(const Val& x, const Val& y, i64 sy, const Node* n) const {
    if (x.k == V_LIN && y.k == V_LIN) return Val::lin(lin_add(x.a, y.a, sy));
    if (x.k == V_QUAD && y.k == V_LIN) { Val r = x; r.c = lin_add(x.c, y.a, sy); return r; }
    if (x.k == V_LIN && y.k == V_QUAD) { Val r; r.k = V_QUAD; r.a = lin_scale(y.a, sy); r.b = y.b; r.c = lin_add(x.a, y.c, sy); return r; }
    err(n, "unsupported operands of + / - (sum of two products, or an array)");
  }
  Val v_mul(const Val& x, const Val& y, const Node* n) const {
    if (x.k == V_LIN && y.k == V_LIN) {
      if (x.a.is_const()) return Val::lin(lin_scale(y.a, x.a.c0));
      if (y.a.is_const()) return Val::lin(lin_scale(x.a, y.a.c0));
      Val r; r.k = V_QUAD; r.a = x.a; r.b = y.a;
      return r;
    }
    if (x.k == V_QUAD && y.is_const()) { Val r = x; r.a = lin_scale(x.a, y.a.c0); r.c = lin_scale(x.c, y.a.c0); return r; }
    if (y.k == V_QUAD && x.is_const()) return v_mul(y, x, n);
    err(n, "expression of degree 3 or an array operand of *");
  }

  // ---- instances
  Inst* instantiate(const std::string& tname, const std::vector<i64>& args, const std::string& name, Inst* parent, const Node* at) {
    auto it = templates.find(tname);
    if (it == templates.end()) err(at, "unknown template " + tname);
    const Template& T = it->second;
    if (args.size() != T.params.size()) err(at, tname + ": " + std::to_string(args.size()) + " parameters given, " + std::to_string(T.params.size()) + " declared");
    insts.emplace_back();
    Inst* in = &insts.back();
    in->tmpl = &T; in->args = args; in->name = name; in->parent = parent;
    if (parent) parent->subs.push_back(in);
    // declare the input signals (their extents may depend on the parameters and on leading `var`s)
    Frame f{in, {}, {}};
    for (size_t p = 0; p < T.params.size(); ++p) f.vars[T.params[p]] = VarArr{{}, {Val::num(args[p])}};
    for (Node* st : T.body->c) {
      if (st->k == S_VAR) { try { exec(st, f); } catch (Error&) {} }
      else if (st->k == S_SIGNAL && st->s == "in") declare_signals(st, f, true);
    }
    for (auto& s : in->sigs) in->pending += s.size();
    return in;
  }
  void maybe_run(Inst* in, const Node* at) {
    if (!in->ran && in->pending == 0) run(in, at);
  }
  void run(Inst* in, const Node* at) {
    if (in->running) err(at, "recursive component execution");
    in->running = true;
    Frame f{in, {}, {}};
    const Template& T = *in->tmpl;
    for (size_t p = 0; p < T.params.size(); ++p) f.vars[T.params[p]] = VarArr{{}, {Val::num(in->args[p])}};
    exec(T.body, f);
    in->running = false;
    in->ran = true;
  }

  void declare_signals(Node* st, Frame& f, bool prepass) {
    const int kind = st->s == "out" ? 0 : (st->s == "in" ? 1 : 2);
    for (Node* d : st->c) {
      if (f.inst->sig_of.count(d->s)) {
        if (kind == 1 && !prepass) continue;   // inputs were declared when the component was created
        err(d, "signal " + d->s + " declared twice");
      }
      SigArr S;
      S.name = d->s; S.kind = kind;
      for (Node* e : d->c[0]->c) {
        i64 v = const_of(eval(e, f), e, "a signal array extent");
        if (v < 0 || v > (1 << 28)) err(e, "signal array extent out of range");
        S.dims.push_back((u32)v);
      }
      const u32 n = S.size();
      S.val.resize(n); S.set.assign(n, 0); S.gate.assign(n, -1);
      f.inst->sig_of[S.name] = (u32)f.inst->sigs.size();
      f.inst->sigs.push_back(std::move(S));
    }
  }

  // ---- references
  struct Ref {
    int kind = 0;          // 1 var, 2 signal, 3 component
    VarArr* var = nullptr;
    Inst* inst = nullptr;  // signal owner
    u32 sig = 0;
    CompArr* comp = nullptr;
    std::string comp_name;
    std::vector<i64> idx;
  };
  static u32 flat_index(const std::vector<u32>& dims, const std::vector<i64>& idx, size_t n_idx, u32& remaining) {
    u32 flat = 0;
    remaining = 1;
    for (size_t d = 0; d < dims.size(); ++d) {
      if (d < n_idx) flat = flat * dims[d] + (u32)idx[d];
      else { flat *= dims[d]; remaining *= dims[d]; }
    }
    return flat;
  }
  Ref resolve(Node* n, Frame& f) {
    // flatten the accessor chain
    std::vector<Node*> chain;
    Node* p = n;
    while (p->k == N_INDEX || p->k == N_MEMBER) { chain.push_back(p); p = p->c[0]; }
    if (p->k != N_ID) err(n, "not an assignable expression");
    std::reverse(chain.begin(), chain.end());
    Ref r;
    size_t ci = 0;
    auto take_indices = [&](size_t max_n, std::vector<i64>& out) {
      while (ci < chain.size() && chain[ci]->k == N_INDEX && out.size() < max_n) {
        out.push_back(const_of(eval(chain[ci]->c[1], f), chain[ci], "an index"));
        ++ci;
      }
    };
    auto vi = f.vars.find(p->s);
    if (vi != f.vars.end()) {
      r.kind = 1; r.var = &vi->second;
      take_indices(vi->second.dims.size(), r.idx);
      if (ci != chain.size()) err(n, "too many indices for variable " + p->s);
      for (size_t d = 0; d < r.idx.size(); ++d) if (r.idx[d] < 0 || r.idx[d] >= (i64)vi->second.dims[d]) err(n, "index out of bounds for " + p->s);
      return r;
    }
    if (!f.inst) err(n, "unknown identifier " + p->s + " (functions only see their parameters and variables)");
    auto si = f.inst->sig_of.find(p->s);
    if (si != f.inst->sig_of.end()) {
      r.kind = 2; r.inst = f.inst; r.sig = si->second;
      SigArr& S = f.inst->sigs[r.sig];
      take_indices(S.dims.size(), r.idx);
      if (ci != chain.size()) err(n, "too many indices for signal " + p->s);
      for (size_t d = 0; d < r.idx.size(); ++d) if (r.idx[d] < 0 || r.idx[d] >= (i64)S.dims[d]) err(n, "index out of bounds for signal " + p->s);
      return r;
    }
    auto cit = f.inst->comps.find(p->s);
    if (cit != f.inst->comps.end()) {
      CompArr& C = cit->second;
      std::vector<i64> cidx;
      take_indices(C.dims.size(), cidx);
      if (cidx.size() != C.dims.size()) err(n, "component array " + p->s + " needs " + std::to_string(C.dims.size()) + " indices");
      u32 rem, flat = flat_index(C.dims, cidx, cidx.size(), rem);
      for (size_t d = 0; d < cidx.size(); ++d) if (cidx[d] < 0 || cidx[d] >= (i64)C.dims[d]) err(n, "index out of bounds for component " + p->s);
      if (ci == chain.size()) {
        r.kind = 3; r.comp = &C; r.comp_name = p->s; r.idx = cidx;
        return r;
      }
      if (chain[ci]->k != N_MEMBER) err(n, "expected .signal after component " + p->s);
      Inst* in = C.inst[flat];
      if (!in) err(n, "component " + p->s + " used before it is created");
      const std::string& sname = chain[ci]->s;
      ++ci;
      auto s2 = in->sig_of.find(sname);
      if (s2 == in->sig_of.end()) {
        if (!in->ran) err(n, "signal " + sname + " of component " + p->s + " read before the component has run (or it is not an input)");
        err(n, "component " + p->s + " (" + in->tmpl->name + ") has no signal " + sname);
      }
      r.kind = 2; r.inst = in; r.sig = s2->second;
      SigArr& S = in->sigs[r.sig];
      take_indices(S.dims.size(), r.idx);
      if (ci != chain.size()) err(n, "too many indices for " + p->s + "." + sname);
      for (size_t d = 0; d < r.idx.size(); ++d) if (r.idx[d] < 0 || r.idx[d] >= (i64)S.dims[d]) err(n, "index out of bounds for " + p->s + "." + sname);
      return r;
    }
    err(n, "unknown identifier " + p->s);
  }
  Val read_sig_range(Inst* in, SigArr& S, u32 flat, u32 count, size_t dim, const Node* n) {
    if (dim == S.dims.size()) {
      if (!S.set[flat]) err(n, "signal " + path_of(in) + "." + S.name + " read before it is assigned");
      Val v = Val::lin(S.val[flat]);
      v.a.sig = true;
      return v;
    }
    Val v;
    v.k = V_ARR;
    const u32 sub = count / S.dims[dim];
    for (u32 i = 0; i < S.dims[dim]; ++i) v.arr.push_back(read_sig_range(in, S, flat + i * sub, sub, dim + 1, n));
    return v;
  }
  Val read_var_range(VarArr& V, u32 flat, u32 count, size_t dim) {
    if (dim == V.dims.size()) return V.v[flat];
    Val v;
    v.k = V_ARR;
    const u32 sub = count / V.dims[dim];
    for (u32 i = 0; i < V.dims[dim]; ++i) v.arr.push_back(read_var_range(V, flat + i * sub, sub, dim + 1));
    return v;
  }
  Val read(const Ref& r, const Node* n) {
    if (r.kind == 1) {
      u32 rem, flat = flat_index(r.var->dims, r.idx, r.idx.size(), rem);
      Val v = read_var_range(*r.var, flat, rem, r.idx.size());
      if (v.k == V_UNSET) err(n, "variable read before it is assigned");
      return v;
    }
    if (r.kind == 2) {
      SigArr& S = r.inst->sigs[r.sig];
      u32 rem, flat = flat_index(S.dims, r.idx, r.idx.size(), rem);
      return read_sig_range(r.inst, S, flat, rem, r.idx.size(), n);
    }
    err(n, "a component is not a value");
  }

  // ---- gates
  std::string path_of(const Inst* in) const {
    std::string p;
    for (const Inst* q = in; q && q->parent; q = q->parent) p = "." + q->name + p;
    return p;
  }
  void interval_of(const Lin& l, i64& lo, i64& hi) const {
    __int128 a = l.c0, b = l.c0;
    for (auto& t : l.t) {
      i64 slo, shi;
      if (t.first & SRC_INPUT) { slo = in_lo[t.first & 0x3fffffffu]; shi = in_hi[t.first & 0x3fffffffu]; }
      else { slo = gates[t.first].lo; shi = gates[t.first].hi; }
      if (t.second >= 0) { a += (__int128)t.second * slo; b += (__int128)t.second * shi; }
      else { a += (__int128)t.second * shi; b += (__int128)t.second * slo; }
    }
    const __int128 L = (__int128)1 << 60;
    lo = (i64)std::max<__int128>(a, -L);
    hi = (i64)std::min<__int128>(b, L);
  }
  void no_inverse_sources(const Lin& l, const Node* at) const {
    for (auto& t : l.t) if (!(t.first & SRC_INPUT) && gates[t.first].op == G_INV0) err(at, "an inverse hint is used outside the IsZero pattern (in * inv)");
  }
  u32 add_gate(Gate g, const Node* at) {
    g.at = at;
    i64 lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    const int nf = (g.op == G_QUAD || g.op == G_ASSERT) ? 3 : (g.op == G_NEZ ? 2 : 1);
    for (int i = 0; i < nf; ++i) { no_inverse_sources(g.f[i], at); interval_of(g.f[i], lo[i], hi[i]); }
    if (g.op == G_NEZ && !g.f[1].t.empty()) err(at, "internal: indicator gate with a non-constant offset");
    switch (g.op) {
      case G_QUAD: case G_ASSERT: {
        __int128 c[4] = {(__int128)lo[0] * lo[1], (__int128)lo[0] * hi[1], (__int128)hi[0] * lo[1], (__int128)hi[0] * hi[1]};
        __int128 a = std::min(std::min(c[0], c[1]), std::min(c[2], c[3])) + lo[2];
        __int128 b = std::max(std::max(c[0], c[1]), std::max(c[2], c[3])) + hi[2];
        const __int128 L = (__int128)1 << 60;
        g.lo = (i64)std::max<__int128>(a, -L); g.hi = (i64)std::min<__int128>(b, L);
        break;
      }
      case G_INV0: g.lo = lo[0]; g.hi = hi[0]; break;      // interval of the inverted integer
      case G_BIT: g.lo = 0; g.hi = 1; break;
      case G_NEZ: g.lo = std::min<i64>(0, g.k) + lo[1]; g.hi = std::max<i64>(0, g.k) + hi[1]; break;
      default: g.lo = lo[0]; g.hi = hi[0]; break;
    }
    // intervals ignore correlations (a long chain of boolean recurrences widens them without bound): they only
    // classify signals and size the inverse table; the evaluator checks the 31-bit range of every value at run time
    const i64 LIM = ((i64)1 << 30) - 1;
    g.lo = std::max(g.lo, -LIM); g.hi = std::min(g.hi, LIM);
    gates.push_back(std::move(g));
    return (u32)gates.size() - 1;
  }
  static Lin lin_gate(u32 g) { Lin l; l.t.emplace_back(g, 1); return l; }

  // the value a signal takes when assigned `v` with <== (kept signals become gates)
  void assign_signal(Inst* in, u32 sig, u32 flat, const Val& v, const Node* at, Frame& f) {
    SigArr& S = in->sigs[sig];
    if (S.set[flat]) err(at, "signal " + path_of(in) + "." + S.name + " assigned twice");
    if (v.k == V_LIN) {
      if (v.a.t.size() > max_terms) {
        Gate g; g.op = G_LIN; g.f[0] = v.a;
        S.val[flat] = lin_gate(add_gate(std::move(g), at));
      } else S.val[flat] = v.a;
    } else if (v.k == V_QUAD) {
      Gate g;
      g.owner = in; g.sig = sig; g.flat = flat;
      // in * inv patterns: one factor is an inverse hint of L, the other a multiple of L
      const Lin* fa = &v.a; const Lin* fb = &v.b;
      auto inv_of = [&](const Lin& l) -> const Gate* {
        if (l.c0 == 0 && l.t.size() == 1 && !(l.t[0].first & SRC_INPUT) && gates[l.t[0].first].op == G_INV0) return &gates[l.t[0].first];
        return nullptr;
      };
      const Gate* ig = inv_of(*fb);
      if (!ig) { ig = inv_of(*fa); std::swap(fa, fb); }
      if (ig) {
        const i64 ci = fb->t[0].second;
        i64 k;
        if (lin_multiple(*fa, ig->f[0], k)) {
          g.op = G_NEZ; g.k = chk((__int128)k * ci, "in * inv"); g.f[0] = ig->f[0]; g.f[1] = lin_const(v.c.c0);
          if (!v.c.t.empty()) {   // k * (x != 0) + (non-constant): the indicator becomes a temporary
            Gate t; t.op = G_NEZ; t.k = g.k; t.f[0] = g.f[0];
            g.op = G_LIN; g.k = 0; g.f[0] = lin_add(lin_gate(add_gate(std::move(t), at)), v.c);
            g.f[1] = Lin();
          }
        } else err(at, "product with an inverse hint that is not of the form (k * x) * inverse(x)");
      } else {
        g.op = G_QUAD; g.f[0] = v.a; g.f[1] = v.b; g.f[2] = v.c;
        for (int i = 0; i < 3; ++i) if (g.f[i].t.size() > max_terms) { Gate t; t.op = G_LIN; t.f[0] = g.f[i]; g.f[i] = lin_gate(add_gate(std::move(t), at)); }
      }
      const u32 id = add_gate(std::move(g), at);
      S.val[flat] = lin_gate(id);
      S.gate[flat] = (int)id;
    } else err(at, "cannot assign this value to a signal");
    S.set[flat] = 1;
    (void)f;
    if (S.kind == 1 && in->pending > 0) { --in->pending; }
  }
  void assign_signal_range(Inst* in, u32 sig, u32 flat, u32 count, size_t dim, const Val& v, const Node* at, Frame& f) {
    SigArr& S = in->sigs[sig];
    if (dim == S.dims.size()) { assign_signal(in, sig, flat, v, at, f); return; }
    if (v.k != V_ARR || v.arr.size() != S.dims[dim]) err(at, "array shape mismatch in assignment to " + S.name);
    const u32 sub = count / S.dims[dim];
    for (u32 i = 0; i < S.dims[dim]; ++i) assign_signal_range(in, sig, flat + i * sub, sub, dim + 1, v.arr[i], at, f);
  }
  void constrain_assign(const Ref& r, const Val& v, const Node* at, Frame& f) {
    SigArr& S = r.inst->sigs[r.sig];
    if (r.inst != f.inst && S.kind != 1) err(at, "only the inputs of a sub-component can be assigned");
    u32 rem, flat = flat_index(S.dims, r.idx, r.idx.size(), rem);
    assign_signal_range(r.inst, r.sig, flat, rem, r.idx.size(), v, at, f);
    if (r.inst != f.inst) maybe_run(r.inst, at);
  }
  // `lhs <-- hint`
  void hint_assign(const Ref& r, Node* rhs, const Node* at, Frame& f) {
    SigArr& S = r.inst->sigs[r.sig];
    if (r.idx.size() != S.dims.size()) err(at, "hints assign one signal at a time");
    u32 rem, flat = flat_index(S.dims, r.idx, r.idx.size(), rem);
    if (S.set[flat]) err(at, "signal " + S.name + " assigned twice");
    Gate g;
    g.owner = r.inst; g.sig = r.sig; g.flat = flat;
    Node* e = rhs;
    // (x >> k) & 1
    auto strip = [](Node* n) { return n; };
    (void)strip;
    if (e->k == N_BIN && e->s == "&" && e->c[1]->k == N_NUM && e->c[1]->n == 1 && e->c[0]->k == N_BIN && e->c[0]->s == ">>") {
      Val x = eval(e->c[0]->c[0], f);
      Val k = eval(e->c[0]->c[1], f);
      if (x.k != V_LIN) err(at, "the operand of a bit-extraction hint must be linear");
      g.op = G_BIT; g.f[0] = x.a; g.k = const_of(k, at, "the shift of a bit-extraction hint");
      if (g.k < 0 || g.k > 30) err(at, "bit index out of range");
    } else if (e->k == N_TERN && e->c[0]->k == N_BIN && e->c[0]->s == "!=" && e->c[0]->c[1]->k == N_NUM && e->c[0]->c[1]->n == 0 &&
               e->c[1]->k == N_BIN && e->c[1]->s == "/" && e->c[1]->c[0]->k == N_NUM && e->c[1]->c[0]->n == 1 &&
               e->c[2]->k == N_NUM && e->c[2]->n == 0) {
      Val x = eval(e->c[0]->c[0], f);
      Val y = eval(e->c[1]->c[1], f);
      if (x.k != V_LIN || y.k != V_LIN || !lin_eq(x.a, y.a)) err(at, "inverse hint must have the shape x != 0 ? 1/x : 0");
      g.op = G_INV0; g.f[0] = x.a;
    } else err(at, "unsupported hint: only (x >> k) & 1 and x != 0 ? 1/x : 0 are known");
    const u32 id = add_gate(std::move(g), at);
    S.val[flat] = lin_gate(id);
    S.gate[flat] = (int)id;
    S.set[flat] = 1;
  }
  // sum_k 2^k * bit_k(F) - F == 0 with bit gates of one form F covering bits 0..n-1 and F provably in [0, 2^n):
  // Num2Bits' closing constraint, true for every input
  bool bit_sum_holds(const Lin& d) const {
    const Lin* F = nullptr;
    Lin rest = d;
    rest.t.clear();
    unsigned long long seen = 0;
    u32 nbits = 0;
    for (auto& t : d.t) {
      const bool is_bit = !(t.first & SRC_INPUT) && gates[t.first].op == G_BIT;
      if (!is_bit) { rest.t.push_back(t); continue; }
      const Gate& g = gates[t.first];
      if (!F) F = &g.f[0];
      else if (!lin_eq(*F, g.f[0])) return false;
      if (g.k < 0 || g.k > 40 || t.second != ((i64)1 << g.k) || (seen >> g.k & 1)) return false;
      seen |= 1ull << g.k;
      ++nbits;
    }
    if (!F || seen != (nbits >= 64 ? ~0ull : (1ull << nbits) - 1)) return false;
    if (!lin_eq(rest, lin_scale(*F, -1))) return false;
    i64 lo, hi;
    interval_of(*F, lo, hi);
    return lo >= 0 && hi < ((i64)1 << nbits);
  }
  void constrain(const Val& l, const Val& r, const Node* at) {
    Val d = v_add(l, r, -1, at);
    Gate g;
    g.op = G_ASSERT;
    if (d.k == V_LIN) {
      if (d.a.t.empty()) { if (d.a.c0 != 0) err(at, "constraint between constants does not hold"); return; }
      if (bit_sum_holds(d.a)) return;
      g.f[2] = d.a;
      if (g.f[2].t.size() > max_terms) { Gate t; t.op = G_LIN; t.f[0] = g.f[2]; g.f[2] = lin_gate(add_gate(std::move(t), at)); }
    } else if (d.k == V_QUAD) {
      // bit * (bit - 1) and in * IsZero(in).out hold by construction
      auto single = [&](const Lin& x, u32& gid) { if (x.c0 == 0 && x.t.size() == 1 && x.t[0].second == 1 && !(x.t[0].first & SRC_INPUT)) { gid = x.t[0].first; return true; } return false; };
      u32 ga;
      if (d.c.t.empty() && d.c.c0 == 0) {
        if (single(d.a, ga) && gates[ga].op == G_BIT && lin_eq(d.b, lin_add(d.a, lin_const(-1)))) return;
        for (int sw = 0; sw < 2; ++sw) {
          const Lin& x = sw ? d.b : d.a; const Lin& y = sw ? d.a : d.b;
          u32 gy;
          i64 k;
          if (single(y, gy) && gates[gy].op == G_NEZ && gates[gy].k == -1 && gates[gy].f[1].t.empty() && gates[gy].f[1].c0 == 1 &&
              lin_multiple(x, gates[gy].f[0], k)) return;
        }
      }
      g.f[0] = d.a; g.f[1] = d.b; g.f[2] = d.c;
    } else err(at, "unsupported constraint");
    add_gate(std::move(g), at);
  }

  // ---- expressions
  Val eval(Node* n, Frame& f) {
    switch (n->k) {
      case N_NUM:
        if (n->big) err(n, "integer literal too large for this loader");
        return Val::num(n->n);
      case N_ID: case N_INDEX: case N_MEMBER: return read(resolve(n, f), n);
      case N_ARR: { Val v; v.k = V_ARR; for (Node* e : n->c) v.arr.push_back(eval(e, f)); return v; }
      case N_UN: {
        Val x = eval(n->c[0], f);
        if (n->s == "-") {
          if (x.k == V_LIN) return Val::lin(lin_scale(x.a, -1));
          if (x.k == V_QUAD) { Val r = x; r.a = lin_scale(x.a, -1); r.c = lin_scale(x.c, -1); return r; }
          err(n, "unsupported operand of unary -");
        }
        if (n->s == "!") return Val::num(const_of(x, n, "the operand of !") == 0);
        err(n, "unsupported unary operator " + n->s);
      }
      case N_TERN: {
        const i64 c = const_of(eval(n->c[0], f), n, "the condition of ?:");
        return eval(c ? n->c[1] : n->c[2], f);
      }
      case N_BIN: {
        const std::string& op = n->s;
        if (op == "&&") { if (!const_of(eval(n->c[0], f), n, "an operand of &&")) return Val::num(0); return Val::num(const_of(eval(n->c[1], f), n, "an operand of &&") != 0); }
        if (op == "||") { if (const_of(eval(n->c[0], f), n, "an operand of ||")) return Val::num(1); return Val::num(const_of(eval(n->c[1], f), n, "an operand of ||") != 0); }
        Val x = eval(n->c[0], f), y = eval(n->c[1], f);
        if (op == "+") return v_add(x, y, 1, n);
        if (op == "-") return v_add(x, y, -1, n);
        if (op == "*") return v_mul(x, y, n);
        const i64 a = const_of(x, n, ("the left operand of " + op).c_str()), b = const_of(y, n, ("the right operand of " + op).c_str());
        if (op == "==") return Val::num(a == b);
        if (op == "!=") return Val::num(a != b);
        if (op == "<") return Val::num(a < b);
        if (op == ">") return Val::num(a > b);
        if (op == "<=") return Val::num(a <= b);
        if (op == ">=") return Val::num(a >= b);
        if (op == "\\") { if (b == 0 || a < 0 || b < 0) err(n, "integer division needs non-negative operands"); return Val::num(a / b); }
        if (op == "/") { if (b == 0 || a % b) err(n, "field division of constants that is not exact"); return Val::num(a / b); }
        if (op == "%") { if (b <= 0 || a < 0) err(n, "% needs non-negative operands"); return Val::num(a % b); }
        if (op == "<<") { if (b < 0 || b > 39 || a < 0) err(n, "shift out of range"); return Val::num(chk((__int128)a << b, "a shift")); }
        if (op == ">>") { if (b < 0 || a < 0) err(n, "shift out of range"); return Val::num(b > 62 ? 0 : a >> b); }
        if (op == "&") { if (a < 0 || b < 0) err(n, "& needs non-negative operands"); return Val::num(a & b); }
        if (op == "|") { if (a < 0 || b < 0) err(n, "| needs non-negative operands"); return Val::num(a | b); }
        if (op == "^") { if (a < 0 || b < 0) err(n, "^ needs non-negative operands"); return Val::num(a ^ b); }
        if (op == "**") { if (b < 0 || b > 62) err(n, "exponent out of range"); __int128 r = 1; for (i64 i = 0; i < b; ++i) { r *= a; chk(r, "a power"); } return Val::num((i64)r); }
        err(n, "unsupported operator " + op);
      }
      case N_CALL: {
        if (templates.count(n->s)) err(n, "a template instantiation is only valid on the right of `component x =`");
        auto fi = functions.find(n->s);
        if (fi == functions.end()) err(n, "unknown function " + n->s);
        const Template& F = fi->second;
        if (n->c.size() != F.params.size()) err(n, n->s + ": wrong number of arguments");
        Frame ff{nullptr, {}, {}};
        for (size_t p = 0; p < F.params.size(); ++p) {
          Val a = eval(n->c[p], f);
          if (a.k == V_QUAD) err(n, "function argument of degree 2");
          ff.vars[F.params[p]] = VarArr{{}, {a}};
        }
        if (++call_depth > 64) err(n, "function recursion too deep");
        Val ret;
        try { exec(F.body, ff); --call_depth; err(n, "function " + n->s + " ended without return"); }
        catch (ReturnValue& r) { --call_depth; ret = r.v; }
        return ret;
      }
      case N_ANON: return eval_anon(n, f);
      default: err(n, "not an expression");
    }
  }
  Val eval_anon(Node* n, Frame& f) {
    std::vector<i64> args;
    for (Node* a : n->c[0]->c) args.push_back(const_of(eval(a, f), a, "a template parameter"));
    std::vector<Val> ins;
    for (Node* a : n->c[1]->c) ins.push_back(eval(a, f));
    std::string name = n->s + "_" + std::to_string(n->line) + "_" + std::to_string(n->pos);
    if (!f.loops.empty()) name += "[" + std::to_string(f.loops.back()) + "]";
    Inst* in = instantiate(n->s, args, name, f.inst, n);
    std::vector<u32> inputs;
    for (u32 s = 0; s < in->sigs.size(); ++s) if (in->sigs[s].kind == 1) inputs.push_back(s);
    if (inputs.size() != ins.size()) err(n, n->s + ": " + std::to_string(ins.size()) + " inputs given, the template declares " + std::to_string(inputs.size()));
    for (size_t k = 0; k < inputs.size(); ++k) assign_signal_range(in, inputs[k], 0, in->sigs[inputs[k]].size(), 0, ins[k], n, f);
    maybe_run(in, n);
    if (!in->ran) err(n, "anonymous component did not receive all its inputs");
    std::vector<Val> outs;
    for (auto& S : in->sigs) if (S.kind == 0) outs.push_back(read_sig_range(in, S, 0, S.size(), 0, n));
    if (outs.size() == 1) return outs[0];
    Val v; v.k = V_ARR; v.arr = outs;
    return v;
  }

  // ---- statements
  void declare_var(Node* d, Frame& f) {
    VarArr V;
    for (Node* e : d->c[0]->c) {
      i64 v = const_of(eval(e, f), e, "a variable array extent");
      if (v < 0 || v > (1 << 24)) err(e, "variable array extent out of range");
      V.dims.push_back((u32)v);
    }
    u32 n = 1;
    for (u32 x : V.dims) n *= x;
    V.v.assign(n, V.dims.empty() ? Val() : Val::num(0));
    if (d->c.size() > 1) {
      Val init = eval(d->c[1], f);
      if (V.dims.empty()) V.v[0] = init;
      else {
        std::function<void(const Val&, u32, u32, size_t)> fill = [&](const Val& v, u32 flat, u32 count, size_t dim) {
          if (dim == V.dims.size()) { V.v[flat] = v; return; }
          if (v.k != V_ARR || v.arr.size() != V.dims[dim]) err(d, "array shape mismatch in the initialiser of " + d->s);
          const u32 sub = count / V.dims[dim];
          for (u32 i = 0; i < V.dims[dim]; ++i) fill(v.arr[i], flat + i * sub, sub, dim + 1);
        };
        fill(init, 0, n, 0);
      }
    } else if (V.dims.empty()) V.v[0] = Val::num(0);
    f.vars[d->s] = std::move(V);
  }
  void create_component(Frame& f, CompArr& C, const std::string& cname, const std::vector<i64>& idx, Node* rhs) {
    if (rhs->k != N_CALL || !templates.count(rhs->s)) err(rhs, "expected a template instantiation");
    std::vector<i64> args;
    for (Node* a : rhs->c) args.push_back(const_of(eval(a, f), a, "a template parameter"));
    u32 rem, flat = flat_index(C.dims, idx, idx.size(), rem);
    if (C.inst[flat]) err(rhs, "component " + cname + " created twice");
    std::string name = cname;
    for (i64 i : idx) name += "[" + std::to_string(i) + "]";
    Inst* in = instantiate(rhs->s, args, name, f.inst, rhs);
    C.inst[flat] = in;
    maybe_run(in, rhs);
  }
  void exec(Node* n, Frame& f) {
    switch (n->k) {
      case S_NOP: case S_LOG: return;
      case S_BLOCK: for (Node* s : n->c) exec(s, f); return;
      case S_SIGNAL:
        if (!f.inst) err(n, "signal declared inside a function");
        declare_signals(n, f, false);
        for (Node* d : n->c) if (d->c.size() > 1) {
          Ref r; r.kind = 2; r.inst = f.inst; r.sig = f.inst->sig_of[d->s];
          if (d->c[2]->s == "<--") hint_assign(r, d->c[1], d, f);
          else constrain_assign(r, eval(d->c[1], f), d, f);
        }
        return;
      case S_VAR: for (Node* d : n->c) declare_var(d, f); return;
      case S_COMP:
        if (!f.inst) err(n, "component declared inside a function");
        for (Node* d : n->c) {
          if (f.inst->comps.count(d->s) || f.inst->sig_of.count(d->s)) err(d, d->s + " declared twice");
          CompArr C;
          for (Node* e : d->c[0]->c) {
            i64 v = const_of(eval(e, f), e, "a component array extent");
            if (v < 0 || v > (1 << 26)) err(e, "component array extent out of range");
            C.dims.push_back((u32)v);
          }
          u32 cnt = 1;
          for (u32 x : C.dims) cnt *= x;
          C.inst.assign(cnt, nullptr);
          f.inst->comps[d->s] = std::move(C);
          if (d->c.size() > 1) create_component(f, f.inst->comps[d->s], d->s, {}, d->c[1]);
        }
        return;
      case S_IF: {
        const i64 c = const_of(eval(n->c[0], f), n, "the condition of if");
        if (c) exec(n->c[1], f);
        else if (n->c.size() > 2) exec(n->c[2], f);
        return;
      }
      case S_FOR: {
        exec(n->c[0], f);
        f.loops.push_back(0);
        u64 guard = 0;
        while (const_of(eval(n->c[1], f), n, "the loop condition")) {
          exec(n->c[3], f);
          exec(n->c[2], f);
          ++f.loops.back();
          if (++guard > (1ull << 28)) err(n, "loop does not terminate");
        }
        f.loops.pop_back();
        return;
      }
      case S_WHILE: {
        f.loops.push_back(0);
        u64 guard = 0;
        while (const_of(eval(n->c[0], f), n, "the loop condition")) {
          exec(n->c[1], f);
          ++f.loops.back();
          if (++guard > (1ull << 28)) err(n, "loop does not terminate");
        }
        f.loops.pop_back();
        return;
      }
      case S_ASSERT: if (!const_of(eval(n->c[0], f), n, "an assert")) err(n, "assert failed while loading the template"); return;
      case S_RET:
        if (f.inst) err(n, "return outside a function");
        throw ReturnValue{eval(n->c[0], f)};
      case S_CONSTR:
        if (!f.inst) err(n, "constraint inside a function");
        constrain(eval(n->c[0], f), eval(n->c[1], f), n);
        return;
      case S_ASSIGN: {
        Ref r = resolve(n->c[0], f);
        const std::string& op = n->s;
        if (r.kind == 3) {
          if (op != "=") err(n, "components are created with =");
          create_component(f, *r.comp, r.comp_name, r.idx, n->c[1]);
          return;
        }
        if (r.kind == 2) {
          if (op == "<==") constrain_assign(r, eval(n->c[1], f), n, f);
          else if (op == "<--") {
            hint_assign(r, n->c[1], n, f);
            if (r.inst != f.inst) { SigArr& S = r.inst->sigs[r.sig]; if (S.kind == 1 && r.inst->pending) { --r.inst->pending; maybe_run(r.inst, n); } }
          }
          else err(n, "signals are assigned with <== or <--");
          return;
        }
        // variable
        if (r.idx.size() != r.var->dims.size()) {
          if (op != "=") err(n, "compound assignment to an array");
          Val v = eval(n->c[1], f);
          u32 rem, flat = flat_index(r.var->dims, r.idx, r.idx.size(), rem);
          std::function<void(const Val&, u32, u32, size_t)> fill = [&](const Val& x, u32 fl, u32 count, size_t dim) {
            if (dim == r.var->dims.size()) { r.var->v[fl] = x; return; }
            if (x.k != V_ARR || x.arr.size() != r.var->dims[dim]) err(n, "array shape mismatch");
            const u32 sub = count / r.var->dims[dim];
            for (u32 i = 0; i < r.var->dims[dim]; ++i) fill(x.arr[i], fl + i * sub, sub, dim + 1);
          };
          fill(v, flat, rem, r.idx.size());
          return;
        }
        u32 rem, flat = flat_index(r.var->dims, r.idx, r.idx.size(), rem);
        Val& dst = r.var->v[flat];
        Val v = eval(n->c[1], f);
        if (op == "=") { dst = v; return; }
        if (dst.k == V_UNSET) err(n, "variable used before it is assigned");
        if (op == "+=") { dst = v_add(dst, v, 1, n); return; }
        if (op == "-=") { dst = v_add(dst, v, -1, n); return; }
        if (op == "*=") { dst = v_mul(dst, v, n); return; }
        Node tmp;
        tmp.k = N_BIN; tmp.s = op.substr(0, op.size() - 1); tmp.line = n->line; tmp.file = n->file;
        Node l, rr;
        l.k = N_NUM; l.n = const_of(dst, n, "the variable"); rr.k = N_NUM; rr.n = const_of(v, n, "the operand");
        tmp.c = {&l, &rr};
        dst = eval(&tmp, f);
        return;
      }
      default: err(n, "unsupported statement");
    }
  }

  // ------------------------------------------------------------------------------------------------ lowering
  void layout_walk(Inst* in, u32& next, std::vector<std::string>* names) {
    const std::string p = path_of(in);
    for (int kind = 0; kind < 3; ++kind)
      for (auto& S : in->sigs) if (S.kind == kind)
        for (u32 i = 0; i < S.size(); ++i) if (S.gate[i] >= 0) {
          gates[S.gate[i]].slot = next++;
          if (names) names->push_back(p + "." + S.name + (S.dims.empty() ? std::string() : "[" + std::to_string(i) + "]"));
        }
    for (Inst* s : in->subs) layout_walk(s, next, names);
  }

  void build(const std::string& path, const std::string& tname, const std::vector<i64>& args, Net& net) {
    std::string text;
    if (!read_file(path, text)) fail("cannot read " + path);
    included.insert(base_of(path) + "#" + std::to_string(text.size()));
    load_text(text, path, dir_of(path));
    auto it = templates.find(tname);
    if (it == templates.end()) fail(path + ": no template named " + tname);
    Inst* top = instantiate(tname, args, "", nullptr, it->second.body);
    // the message bytes
    std::vector<u32> inputs;
    for (u32 s = 0; s < top->sigs.size(); ++s) if (top->sigs[s].kind == 1) inputs.push_back(s);
    if (inputs.size() != 1 || top->sigs[inputs[0]].dims.size() != 1) fail(tname + ": expected exactly one input signal array (the message bytes)");
    SigArr& M = top->sigs[inputs[0]];
    n_in = M.size();
    in_lo.assign(n_in, 0); in_hi.assign(n_in, 255);
    for (u32 i = 0; i < n_in; ++i) { M.val[i].t.emplace_back(SRC_INPUT | i, 1); M.set[i] = 1; }
    top->pending = 0;
    run(top, it->second.body);
    // outputs
    std::vector<const SigArr*> outs;
    for (auto& S : top->sigs) if (S.kind == 0) outs.push_back(&S);
    const SigArr* o_match = nullptr; const SigArr* o_reveal = nullptr;
    for (const SigArr* S : outs) { if (S->dims.empty() && !o_match) o_match = S; else if (S->dims.size() == 1 && !o_reveal) o_reveal = S; }
    if (!o_match || !o_reveal || o_reveal->size() != n_in) fail(tname + ": expected a scalar output (match) and an output array of the message length (reveal)");
    std::vector<Lin> out_forms;
    if (!o_match->set[0]) fail(tname + ": output " + o_match->name + " is not assigned");
    out_forms.push_back(o_match->val[0]);
    for (u32 i = 0; i < n_in; ++i) { if (!o_reveal->set[i]) fail(tname + ": output " + o_reveal->name + " is not fully assigned"); out_forms.push_back(o_reveal->val[i]); }
    for (u32 i = 0; i < out_forms.size(); ++i) { Gate g; g.op = G_OUT; g.f[0] = out_forms[i]; g.out_index = i; add_gate(std::move(g), it->second.body); }
    // every component must have run
    for (auto& in : insts) if (!in.ran) fail("component " + path_of(&in) + " (" + in.tmpl->name + ") never received all its inputs");

    // layout: kept signals in the compiler's numbering order (outputs, inputs, intermediates; sub-components in creation order)
    u32 next = 0;
    net.names.clear();
    layout_walk(top, next, &net.names);
    net.n_kept = next;
    for (auto& g : gates) if (!g.owner && g.op != G_ASSERT && g.op != G_OUT) g.slot = next++;
    legalize(next);
    net.n_temp = next - net.n_kept;
    net.n_in = n_in;
    net.n_out = (u32)out_forms.size();
    net.boolean.assign(net.n_kept, 0);
    for (auto& g : gates) {
      if (g.op == G_INV0) net.inv_need = std::max<u32>(net.inv_need, (u32)std::min<i64>(std::max<i64>(std::llabs(g.lo), std::llabs(g.hi)), 1 << 16));
      if (g.slot < net.n_kept && g.op != G_INV0 && g.lo >= 0 && g.hi <= 1) net.boolean[g.slot] = 1;
    }
    if (next >= 0x1fffffffu) fail("the circuit is too large");
    localize(net);
    emit(net);
    finish_region(net);
  }

  // the region's slot descriptors -> periodic runs + transposed tables (Net::runs / pd / tabs)
  static void finish_region(Net& net) {
    const std::vector<u32>& d = net.slot_desc;
    const u32 n = net.n_kept;
    net.runs.clear(); net.pd.clear(); net.tabs.clear();
    // tables, transposed; a table-served word is decoded without a load (zkwg_expand_dec.h ZkDecNetP: value, table inverse or r - m
    // with m < 2^28), so a negative stored word must be small
    auto check = [&](const std::vector<u32>& t) {
      for (u32 w : t) {
        if (w & VAL_INVERSE) continue;
        const int v = (int)(w << 1) >> 1;
        if (v <= -(1 << 28)) fail("a table-served signal of the regex template holds a value below -2^28");
      }
    };
    check(net.fn_tab); check(net.chain.tab); check(net.bchain.tab);
    // descriptor of slot r as (type | column, position)
    auto entry = [&](u32 r, u32& e0, u32& e1) {
      const u32 x = r < d.size() ? d[r] : 0u, t = x >> 29;
      if (t < 4u) { e0 = 0; e1 = 0; return; }
      const u32 ty = t == 7u ? 3u : (t == 6u ? 2u : 1u);
      e0 = (ty << 30) | ((x >> 16) & 0x1fffu); e1 = x & 0xffffu;
    };
    // slot b continues slot a one period later: the same function, one position on (an evaluated slot continues as an evaluated one)
    auto follows = [&](u32 a, u32 b) {
      u32 a0, a1, b0, b1;
      entry(a, a0, a1); entry(b, b0, b1);
      return a0 == b0 && ((a0 >> 30) == 0u || b1 == a1 + 1u);
    };
    const u32 PMAX = 8192;
    u32 r = 0;
    while (r < n) {
      // the period that carries furthest from r
      u32 bestP = 0, best_len = 0;
      for (u32 P = 1; P <= PMAX && r + P < n; ++P) {
        if (!follows(r, r + P)) continue;
        u32 len = 0;
        while (r + len + P < n && follows(r + len, r + len + P)) ++len;
        if (len >= 2 * P && len + P > best_len) { best_len = len + P; bestP = P; }
        if (bestP && best_len >= 8 * bestP) break;
      }
      Net::Run R;
      R.start = r; R.pd0 = (u32)(net.pd.size() / 2);
      if (bestP) { R.period = bestP; R.nslots = best_len; }
      else {
        // irregular stretch: up to the next slot where a period starts (one run of one period)
        u32 e = r + 1;
        for (; e < n; ++e) {
          bool per = false;
          for (u32 P = 1; P <= 512 && e + 3 * P < n && !per; ++P) {
            u32 len = 0;
            while (len < 2 * P && follows(e + len, e + len + P)) ++len;
            per = len >= 2 * P;
          }
          if (per || e - r >= PMAX) break;
        }
        R.period = e - r; R.nslots = e - r;
      }
      // positions relative to the run's most frequent one, so that zk_expand can ask for the position word before it has the descriptor
      {
        std::map<u32, u32> freq;
        for (u32 q = 0; q < R.period; ++q) { u32 e0, e1; entry(r + q, e0, e1); if (e0 >> 30) ++freq[e1]; }
        R.pos0 = 0;
        u32 best = 0;
        for (auto& kv : freq) if (kv.second > best) { best = kv.second; R.pos0 = kv.first; }
      }
      for (u32 q = 0; q < R.period; ++q) { u32 e0, e1; entry(r + q, e0, e1); net.pd.push_back(e0); net.pd.push_back((e0 >> 30) ? e1 - R.pos0 : 0u); }
      net.runs.push_back(R);
      r += R.nslots;
    }
    // Columns.  A table of the loader may serve many slots (the same comparator in several transitions); the transposed tables
    // give every descriptor of the LARGEST run a column of its own, in slot order, so that the lanes of a wavefront -- consecutive
    // slots of one position -- read consecutive words of one row; the other runs reuse those columns where they name the same table.
    {
      const size_t fcells = (size_t)net.chain.smax * 256, bcells = (size_t)net.bchain.smax * net.bchain.fdim * 256;
      std::vector<u32> cols[4];                      // [type] -> loader table of each column
      std::map<u32, u32> col_of[4];                  // [type] loader table -> a column that holds it
      std::vector<size_t> order(net.runs.size());
      for (size_t k = 0; k < order.size(); ++k) order[k] = k;
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return net.runs[x].nslots > net.runs[y].nslots; });
      for (size_t oi = 0; oi < order.size(); ++oi) {
        const Net::Run& R = net.runs[order[oi]];
        for (u32 q = 0; q < R.period; ++q) {
          u32& e0 = net.pd[2 * (R.pd0 + q)];
          const u32 ty = e0 >> 30, old = e0 & 0x3fffffffu;
          if (!ty) continue;
          auto it = col_of[ty].find(old);
          u32 col;
          if (oi == 0 || it == col_of[ty].end()) {
            col = (u32)cols[ty].size();
            cols[ty].push_back(old);
            if (it == col_of[ty].end()) col_of[ty].emplace(old, col);
          } else col = it->second;
          e0 = (ty << 30) | col;
        }
      }
      net.nL = (u32)cols[1].size(); net.nF = (u32)cols[2].size(); net.nB = (u32)cols[3].size();
      net.offF = 256u * net.nL;
      const size_t offB = (size_t)net.offF + fcells * net.nF, total = offB + bcells * net.nB + 1;
      if (total >= (1u << 30)) fail("the regex template's tables are too large");
      net.offB = (u32)offB;
      net.tabs.assign(total, 0);
      for (u32 t = 0; t < net.nL; ++t) for (u32 v = 0; v < 256; ++v) net.tabs[(size_t)v * net.nL + t] = net.fn_tab[(size_t)cols[1][t] * 256 + v];
      for (u32 t = 0; t < net.nF; ++t) for (size_t c = 0; c < fcells; ++c) net.tabs[net.offF + c * net.nF + t] = net.chain.tab[cols[2][t] * fcells + c];
      for (u32 t = 0; t < net.nB; ++t) for (size_t c = 0; c < bcells; ++c) net.tabs[net.offB + c * net.nB + t] = net.bchain.tab[cols[3][t] * bcells + c];
    }
    // Dense tables for the long runs (the per-byte component block of a regex circuit: 96 % of the region).  Rows = (forward state, byte);
    // a byte-local column repeats its 256 words in every state's rows.  rows x period words: 6.5 MB for the stand-in (29 states, period
    // 220), 14 MB at the real circuit's size -- read 256 contiguous bytes per wavefront, and only the rows of (state, byte) pairs that occur.
    {
      const u32 frows = std::max<u32>(net.chain.smax, 1u) * 256u;
      const size_t fcells = (size_t)net.chain.smax * 256;
      for (Net::Run& R : net.runs) {
        R.dense = 0xffffffffu;
        if (R.period < 32 || R.nslots < 64 * R.period || getenv("ZKWG_NET_DENSE_OFF")) continue;
        bool ok = true;
        for (u32 q = 0; q < R.period && ok; ++q) {
          const u32 e0 = net.pd[2 * (R.pd0 + q)], e1 = net.pd[2 * (R.pd0 + q) + 1], ty = e0 >> 30;
          ok = (ty == 1u || ty == 2u) && e1 == 0u;
        }
        const size_t words = (size_t)frows * R.period;
        if (!ok || words * 4 > (64u << 20) || net.tabs.size() + words >= (1u << 30)) continue;
        R.dense = (u32)net.tabs.size();
        net.tabs.resize(net.tabs.size() + words, 0);
        for (u32 q = 0; q < R.period; ++q) {
          const u32 e0 = net.pd[2 * (R.pd0 + q)], ty = e0 >> 30, col = e0 & 0x3fffffffu;
          for (u32 row = 0; row < frows; ++row) {
            const u32 byte = row & 255u, fs = row >> 8;
            u32 w;
            if (ty == 1u) w = net.tabs[(size_t)byte * net.nL + col];
            else w = fs < net.chain.smax ? net.tabs[net.offF + ((size_t)fs * 256 + byte) * net.nF + col] : 0u;
            net.tabs[R.dense + (size_t)row * R.period + q] = w;
          }
        }
        (void)fcells;
      }
    }
    if (getenv("ZKWG_DEBUG_NET")) {
      fprintf(stderr, "[zkwg] region: %u slots in %zu runs, %zu descriptors, tables %u + %u + %u columns (%.1f MB)\n", n, net.runs.size(), net.pd.size() / 2,
              net.nL, net.nF, net.nB, net.tabs.size() * 4 / 1e6);
      for (const Net::Run& R : net.runs) if (R.nslots > 4 * R.period) fprintf(stderr, "[zkwg]   run at %u: period %u x %u\n", R.start, R.period, R.nslots / R.period);
    }
  }

  // Operand limits of a record (zkwg_net_core.h): a product has <= 2 + 2 + 4 terms (A, B, C), every other gate
  // one sum of <= 8 terms; half of that when a coefficient does not fit 16 bits.  Longer sums are cut into partial
  // sums (temporaries).  Terms are grouped by the level at which they become available, so that a partial sum is
  // ready as early as possible (the 27-term sum of a MultiNOR over one byte's transitions costs no extra level).
  void legalize(u32& next_slot) {
    std::vector<Gate> old;
    old.swap(gates);
    std::vector<u32> remap(old.size(), 0), level;
    auto lvl = [&](u32 src) { return (src & SRC_INPUT) ? 0u : level[src]; };
    auto is_wide = [](const Lin& l) { for (auto& t : l.t) if (t.second < -32768 || t.second > 32767) return true; return false; };
    auto push = [&](Gate&& g) {
      u32 lv = 0;
      const int nf = (g.op == G_QUAD || g.op == G_ASSERT) ? 3 : (g.op == G_NEZ ? 2 : 1);
      for (int i = 0; i < nf; ++i) for (auto& t : g.f[i].t) lv = std::max(lv, lvl(t.first));
      gates.push_back(std::move(g));
      level.push_back(lv + 1);
      return (u32)gates.size() - 1;
    };
    std::function<void(Lin&, u32, const Gate&)> shrink = [&](Lin& f, u32 limit, const Gate& like) {
      while (f.t.size() > limit) {
        std::stable_sort(f.t.begin(), f.t.end(), [&](const std::pair<u32, i64>& x, const std::pair<u32, i64>& y) { return lvl(x.first) < lvl(y.first); });
        // one partial sum of the earliest terms
        size_t take = std::min<size_t>(8, f.t.size() - limit + 1);
        Lin part;
        part.t.assign(f.t.begin(), f.t.begin() + take);
        if (is_wide(part) && take > 4) { take = 4; part.t.resize(4); }
        std::sort(part.t.begin(), part.t.end());
        Gate pg;
        pg.op = G_LIN; pg.f[0] = part; pg.at = like.at;
        i64 lo, hi;
        interval_of(part, lo, hi);
        pg.lo = lo; pg.hi = hi;
        pg.slot = next_slot++;
        const u32 id = push(std::move(pg));
        f.t.erase(f.t.begin(), f.t.begin() + take);
        f.t.emplace_back(id, 1);
      }
      std::sort(f.t.begin(), f.t.end());
    };
    for (u32 gi = 0; gi < old.size(); ++gi) {
      Gate g = std::move(old[gi]);
      const int nf = (g.op == G_QUAD || g.op == G_ASSERT) ? 3 : (g.op == G_NEZ ? 2 : 1);
      for (int i = 0; i < nf; ++i) {
        for (auto& t : g.f[i].t) if (!(t.first & SRC_INPUT)) t.first = remap[t.first];
        std::sort(g.f[i].t.begin(), g.f[i].t.end());
      }
      bool wide = false;
      for (int i = 0; i < nf; ++i) wide = wide || is_wide(g.f[i]);
      if (g.op == G_QUAD || g.op == G_ASSERT) {
        shrink(g.f[0], wide ? 1 : 2, g); shrink(g.f[1], wide ? 1 : 2, g); shrink(g.f[2], wide ? 2 : 4, g);
      } else shrink(g.f[0], wide ? 4 : 8, g);
      remap[gi] = push(std::move(g));
    }
  }

  // Byte-local gates leave the evaluator (Net::fn_tab / mask_tab).  support: -1 = no message byte, i >= 0 = byte i only,
  // -2 = several.  A gate with single support is LOCAL unless it is an assertion or an output.  What a per-email gate
  // reads from a local gate: a boolean becomes a FRONTIER bit (one of its byte's mask bits); anything else keeps the
  // local gate (and its cone) in the evaluator.  Local gates are identified up to the byte they look at by a structural
  // signature, so one 256-entry table serves the same comparator at every position.
  void localize(Net& net) {
    const u32 n = (u32)gates.size();
    g_skip.assign(n, 0); g_front.assign(n, -1); g_sup.assign(n, -1); g_chain.assign(n, 0); g_fpos.assign(n, -1);
    net.slot_desc.assign(net.n_kept, 0);
    if (getenv("ZKWG_NET_LOCALIZE") && !atoi(getenv("ZKWG_NET_LOCALIZE"))) return;
    auto nforms = [](const Gate& g) { return (g.op == G_QUAD || g.op == G_ASSERT) ? 3 : (g.op == G_NEZ ? 2 : 1); };
    std::vector<long long> sup(n, -1);
    std::vector<u64> sig(n, 0);
    auto mix = [](u64 h, u64 v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h * 0xff51afd7ed558ccdull; };
    for (u32 g = 0; g < n; ++g) {
      const Gate& G = gates[g];
      long long sp = -1;
      u64 h = mix(G.op, (u64)G.k);
      for (int i = 0; i < nforms(G); ++i) {
        h = mix(h, (u64)G.f[i].c0 + 17u * i);
        for (auto& t : G.f[i].t) {
          const long long ts = (t.first & SRC_INPUT) ? (long long)(t.first & 0x1fffffffu) : sup[t.first];
          if (ts == -2 || (ts >= 0 && sp >= 0 && ts != sp)) sp = -2; else if (ts >= 0 && sp != -2) sp = ts;
          h = mix(mix(h, (u64)t.second), (t.first & SRC_INPUT) ? 0x5bd1e995u : sig[t.first]);
        }
      }
      sup[g] = sp; sig[g] = h;
    }
    g_sup = sup;
    std::vector<u8> local(n, 0), need(n, 0), front(n, 0);
    for (u32 g = 0; g < n; ++g) local[g] = sup[g] >= 0 && gates[g].op != G_ASSERT && gates[g].op != G_OUT;
    // value of a local gate for byte value v (the arithmetic of zk_net_record): memoised over its cone
    std::vector<long long> memo(n, 0); std::vector<u32> stamp(n, 0); u32 now = 0;
    std::function<long long(u32, long long, u32&)> eval = [&](u32 g, long long v, u32& word) -> long long {
      if (stamp[g] == now) { word = 0; return memo[g]; }
      const Gate& G = gates[g];
      long long f[3] = {0, 0, 0};
      for (int i = 0; i < nforms(G); ++i) {
        f[i] = G.f[i].c0;
        for (auto& t : G.f[i].t) { u32 w; f[i] += t.second * ((t.first & SRC_INPUT) ? v : eval(t.first, v, w)); }
      }
      long long r = f[0];
      if (G.op == G_QUAD) r = f[0] * f[1] + f[2];
      else if (G.op == G_NEZ) r = (f[0] != 0 ? G.k : 0) + f[1];
      else if (G.op == G_BIT) r = f[0] < 0 ? -1 : ((f[0] >> G.k) & 1);
      word = G.op == G_INV0 ? (f[0] == 0 ? 0u : (VAL_INVERSE | ((u32)f[0] & 0x7fffffffu))) : ((u32)r & 0x7fffffffu);
      stamp[g] = now; memo[g] = r;
      return r;
    };
    // tables per signature (representative = first gate with it)
    std::map<u64, std::vector<u32>> words_of;       // stored words for v = 0 .. 255
    std::map<u64, bool> plain_of;                   // every value a plain non-negative 30-bit integer or an inverse in range
    auto table = [&](u32 g) -> const std::vector<u32>& {
      auto it = words_of.find(sig[g]);
      if (it != words_of.end()) return it->second;
      std::vector<u32> w(256);
      bool plain = true;
      for (u32 v = 0; v < 256; ++v) {
        ++now;
        u32 word = 0;
        const long long r = eval(g, (long long)v, word);
        // (eval returns word = 0 for a memo hit; the top-level call is never one)
        w[v] = word;
        if (gates[g].op == G_INV0) { const int d = (int)(word << 1) >> 1; if (std::llabs((long long)d) > (long long)net.inv_need) plain = false; }
        else if (r < 0 || r >= (1ll << 30)) plain = false;
      }
      plain_of[sig[g]] = plain;
      return words_of.emplace(sig[g], std::move(w)).first->second;
    };
    // The signature is a 64-bit structural hash, not a proof of equality: every local gate that shares a table with an earlier
    // one is checked against that table on a few byte values (its own cone evaluated), and a template on which two different
    // functions collide is refused rather than given a wrong witness.
    {
      std::map<u64, u32> first_of;
      std::map<u64, u32> sharers;          // per signature: gates checked so far
      static const u32 probes[6] = {0u, 10u, 59u, 61u, 97u, 255u};
      for (u32 g = 0; g < n; ++g) {
        if (!local[g]) continue;
        auto it = first_of.find(sig[g]);
        if (it == first_of.end()) { first_of.emplace(sig[g], g); continue; }
        const std::vector<u32>& w = table(it->second);
        // the first gate that shares a table is compared on all 256 byte values (six probes could miss two functions that
        // differ elsewhere); the later ones on six values that move with the gate, which cover every value between them
        const bool full = sharers[sig[g]]++ == 0;
        for (u32 q = 0; q < (full ? 256u : 6u); ++q) {
          const u32 v = full ? q : (probes[q] + 7u * g) & 255u;
          ++now;
          u32 word = 0;
          eval(g, (long long)v, word);
          if (word != w[v]) fail("two different byte-local functions of the regex template share a structural signature (hash collision)");
        }
      }
    }
    // the recurrences, if the circuit has them, become scans (chain_pass below): their gates leave the list
    std::vector<int> cbit(n, -1), cbit2(n, -1);
    if (chain_pass(net, 0, sup, sig, local, eval, now, cbit, cbit2) && chain_pass(net, 1, sup, sig, local, eval, now, cbit2, cbit)) {
      // forward-chain booleans that only the backward chain read are no longer read by the list: they keep their table, not their bit record
      std::vector<u8> read(n, 0);
      for (u32 g = 0; g < n; ++g) {
        if (local[g] || g_chain[g]) continue;
        for (int i = 0; i < nforms(gates[g]); ++i) for (auto& t : gates[g].f[i].t) if (!(t.first & SRC_INPUT)) read[t.first] = 1;
      }
      for (u32 g = 0; g < n; ++g) if (cbit[g] >= 0 && !read[g]) { cbit[g] = -1; --net.chain.n_front; }
    }
    // needed gates: everything that is neither local nor served from the chain tables; then what they read
    std::vector<u32> work;
    for (u32 g = 0; g < n; ++g) if (!local[g] && !g_chain[g]) { need[g] = 1; work.push_back(g); }
    // local kept gates whose table holds a value zk_expand could not decode stay in the evaluator
    for (u32 g = 0; g < n; ++g)
      if (local[g] && gates[g].slot < net.n_kept) { table(g); if (!plain_of[sig[g]]) { need[g] = 1; work.push_back(g); } }
    while (!work.empty()) {
      const u32 g = work.back(); work.pop_back();
      const Gate& G = gates[g];
      for (int i = 0; i < nforms(G); ++i)
        for (auto& t : G.f[i].t) {
          if (t.first & SRC_INPUT) continue;
          const u32 sgi = t.first;
          if (!local[sgi] || need[sgi] || front[sgi]) continue;
          const Gate& S = gates[sgi];
          if (S.op != G_INV0 && S.lo >= 0 && S.hi <= 1) front[sgi] = 1;
          else { need[sgi] = 1; work.push_back(sgi); }
        }
    }
    // frontier functions -> mask bits
    std::map<u64, u32> bit_of;
    for (u32 g = 0; g < n; ++g) if (front[g] && !bit_of.count(sig[g])) { const u32 b = (u32)bit_of.size(); bit_of[sig[g]] = b; }
    net.mask_words = (u32)((bit_of.size() + ZKC_MASK_BITS - 1) / ZKC_MASK_BITS);
    net.mask_tab.assign((size_t)256 * net.mask_words, 0);
    for (u32 g = 0; g < n; ++g) {
      if (!front[g]) continue;
      const u32 b = bit_of[sig[g]];
      g_front[g] = (int)b; g_fpos[g] = (int)sup[g];
      ++net.n_frontier;
    }
    // chain gates the list reads: bits of the chain mask words, which follow the byte-local ones in every position's mask region
    for (u32 g = 0; g < n; ++g) {
      if (cbit[g] >= 0) g_front[g] = (int)(net.mask_words * ZKC_MASK_BITS) + cbit[g];
      if (cbit2[g] >= 0) g_front[g] = (int)((net.mask_words + net.chain.mask_words) * ZKC_MASK_BITS) + cbit2[g];
    }
    for (auto& kv : bit_of) {
      // a representative of the signature
      u32 rep = 0;
      while (!(front[rep] && sig[rep] == kv.first)) ++rep;
      const std::vector<u32>& w = table(rep);
      for (u32 v = 0; v < 256; ++v) if (w[v] & 1u) net.mask_tab[(size_t)v * net.mask_words + kv.second / ZKC_MASK_BITS] |= 1u << (kv.second % ZKC_MASK_BITS);
    }
    // local gates that are neither needed nor frontier: not evaluated; the kept ones get a function table
    std::map<u64, u32> fn_of;
    for (u32 g = 0; g < n; ++g) {
      if (!local[g] || need[g]) continue;
      if (!front[g]) { g_skip[g] = 1; ++net.n_local; }
      const Gate& G = gates[g];
      if (G.slot >= net.n_kept || front[g]) continue;     // (a frontier gate still writes its word to the image)
      auto it = fn_of.find(sig[g]);
      if (it == fn_of.end()) {
        const std::vector<u32>& w = table(g);
        it = fn_of.emplace(sig[g], (u32)(net.fn_tab.size() / 256)).first;
        net.fn_tab.insert(net.fn_tab.end(), w.begin(), w.end());
      }
      if (it->second >= 0x1fffu || sup[g] >= 0x10000) fail("too many distinct byte-local functions in the regex template");   // (bits 29, 30 of a descriptor tell the chains' slots)
      net.slot_desc[G.slot] = 0x80000000u | (it->second << 16) | (u32)sup[g];
    }
    if (getenv("ZKWG_DEBUG_NET"))
      fprintf(stderr, "[zkwg] byte-local gates: %u of %u leave the evaluator (%zu function tables), %u served from %u mask word(s) per byte (%zu frontier functions)\n",
              net.n_local, n, net.fn_tab.size() / 256, net.n_frontier, net.mask_words, bit_of.size());
  }

  // The recurrences of a regex circuit as scans.
  //
  // Forward pass.  position(g) = the largest message byte g depends on; the gates with several bytes in their support whose position
  // is i form block i.  A block reads its own byte, the byte-local gates of that byte, constants, and a bounded set of values of
  // earlier positions (S_i: states[i][*] of a zk-regex circuit) -- so every gate of the block is a function of (valuation of S_i,
  // byte i), and the valuation of S_{i+1} another.  The reachable valuations are enumerated from the (empty) S_0 over all 256 byte
  // values and numbered (the state), and the block is tabulated: delta (next state), the stored words of its kept gates, the mask
  // bits of the boolean ones the list still reads.  Positions with the same block descriptor (gates, operands named by their role,
  // byte-local operands by signature) share a class of tables -- a periodic circuit is tabulated at a handful of positions.  Chains
  // that run backwards over the message land in the block of the last byte (their support is everything); that block and all
  // after the first oversized one are left to
  //
  // the backward pass: the same construction on the mirrored message over what is left of the list, whose leaves now include the
  // forward chain's booleans -- the symbol a position contributes is (forward state entering it, its byte), so the tables get one
  // more dimension.  (zk-regex's is_consecutive chain, the stand-in's `live` chain.)
  //
  // Anything the scheme cannot express (a non-boolean chain value read by the list, more than 255 states, values outside the stored
  // range) makes a pass return false with nothing changed: the list then evaluates those gates as before.
  typedef std::function<long long(u32, long long, u32&)> LocalEval;
  // Whether a chain value the list reads is boolean is first taken on trust (value intervals cannot prove it for a recurrence: they
  // widen at every step) and verified on the tabulated values -- a gate found otherwise joins `nonbool` and the pass is repeated;
  // when that makes the state space explode (a counter taken for a state), the pass falls back to what the intervals prove.
  bool chain_pass(Net& net, int pass, const std::vector<long long>& sup, const std::vector<u64>& sig, const std::vector<u8>& local,
                  LocalEval& eval, u32& now, std::vector<int>& cbit, const std::vector<int>& cbit_fwd) {
    if (chain_limit >= 0 ? chain_limit <= pass : (getenv("ZKWG_NET_CHAIN") && atoi(getenv("ZKWG_NET_CHAIN")) <= pass)) return false;
    std::vector<u8> nonbool(gates.size(), 0);
    for (int round = 0; round < 6; ++round) {
      const int r = chain_try(net, pass, /*trust=*/true, nonbool, sup, sig, local, eval, now, cbit, cbit_fwd);
      if (r == 1) return true;
      if (r == 0) break;
    }
    std::fill(nonbool.begin(), nonbool.end(), 0);
    return chain_try(net, pass, /*trust=*/false, nonbool, sup, sig, local, eval, now, cbit, cbit_fwd) == 1;
  }
  // -> 1 collapsed, 0 not (nothing changed), 2 `nonbool` grew: try again
  int chain_try(Net& net, int pass, bool trust, std::vector<u8>& nonbool, const std::vector<long long>& sup, const std::vector<u64>& sig,
                const std::vector<u8>& local, LocalEval& eval, u32& now, std::vector<int>& cbit, const std::vector<int>& cbit_fwd) {
    const u32 n = (u32)gates.size(), N = n_in;
    const bool dbg = getenv("ZKWG_DEBUG_NET") != nullptr, bwd = pass == 1;
    const char* tag = bwd ? "backward chain" : "chain";
    ChainTab& T = bwd ? net.bchain : net.chain;
    const ChainTab& F = net.chain;                      // (backward pass: the forward tables supply the leaves' values)
    const u32 fdim = bwd ? std::max<u32>(F.smax, 1) : 1u;
    auto nforms = [](const Gate& g) { return (g.op == G_QUAD || g.op == G_ASSERT) ? 3 : (g.op == G_NEZ ? 2 : 1); };
    auto give_up = [&](const char* why) { if (dbg) fprintf(stderr, "[zkwg] %s: not collapsed%s (%s)\n", tag, trust ? " on trust" : "", why); return 0; };
    if (N < 16) return give_up("short message");
    if (bwd && !F.end) return give_up("no forward chain");
    auto tp = [&](int p) { return bwd ? (int)N - 1 - p : p; };      // position in walking order <-> message byte
    // leaves: byte-local gates, and in the backward pass the forward chain's mask bits (functions of the position's symbol)
    std::vector<u8> leaf(n, 0);
    std::vector<int> pos(n, -1);
    for (u32 g = 0; g < n; ++g) {
      if (local[g]) { leaf[g] = 1; pos[g] = tp((int)sup[g]); continue; }
      if (bwd && g_chain[g]) { if (cbit_fwd[g] >= 0) { leaf[g] = 2; pos[g] = tp(g_fpos[g]); } continue; }
      if (sup[g] != -2) continue;
      int p = -1;
      for (int i = 0; i < nforms(gates[g]); ++i)
        for (auto& t : gates[g].f[i].t) p = std::max(p, (t.first & SRC_INPUT) ? tp((int)(t.first & 0x1fffffffu)) : pos[t.first]);
      pos[g] = p;
    }
    std::vector<u8> cand(n, 0);
    for (u32 g = 0; g < n; ++g)
      if (sup[g] == -2 && !g_chain[g] && gates[g].op != G_ASSERT && gates[g].op != G_OUT && pos[g] >= 0 && pos[g] < (int)N) cand[g] = 1;
    // Which candidates really are chain gates.  A gate of the list may read the chain only through booleans (mask bits); a chain gate
    // may not read the list, nor a raw byte of another position, nor a value from more than ZKC_CHAIN_REACH positions back (bounded
    // memory: what runs against the walking direction sits at the last position and reads the whole message): a candidate on the
    // wrong side of a rule joins the list (the partial sums of a MultiOR over every position's accept state, for instance -- a
    // counter, not a state), until nothing changes.  Operands precede their readers in the gate order, so a descending pass
    // settles the first rule and an ascending one the others.  Then the positions are cut at the first block that is still far
    // larger than the typical one.
    auto boolean = [&](u32 o) { const Gate& O = gates[o]; return O.op != G_INV0 && !nonbool[o] && (trust || (O.lo >= 0 && O.hi <= 1)); };
    auto in_list = [&](u32 g) { return !local[g] && !cand[g] && !g_chain[g] && sup[g] != -1; };
    const int ZKC_CHAIN_REACH = 4;
    u32 end = N;
    for (;;) {
      for (bool changed = true; changed;) {
        changed = false;
        for (u32 g = n; g-- > 0;) {
          if (!in_list(g)) continue;
          for (int i = 0; i < nforms(gates[g]); ++i)
            for (auto& t : gates[g].f[i].t)
              if (!(t.first & SRC_INPUT) && cand[t.first] && !boolean(t.first)) { cand[t.first] = 0; changed = true; }
        }
        for (u32 g = 0; g < n; ++g) {
          if (!cand[g]) continue;
          for (int i = 0; i < nforms(gates[g]) && cand[g]; ++i)
            for (auto& t : gates[g].f[i].t) {
              bool bad;
              if (t.first & SRC_INPUT) bad = tp((int)(t.first & 0x1fffffffu)) != pos[g];
              else if (sup[t.first] == -1) bad = false;
              else {
                const int po = pos[t.first], pg = pos[g];
                bad = (!cand[t.first] && !leaf[t.first]) || (po < pg - ZKC_CHAIN_REACH && !(leaf[t.first] == 2 && po == 0))   // (the mirrored
                      // walk starts where the forward pass took the first steps of this chain: their results may be read a little further on)
                      // what runs the other way starts at the last position, and its first steps are within reach: there a byte-local
                      // value of an earlier position is not taken (a forward gate has no use for one), so that this pass does not
                      // swallow the head of the other pass's chain
                      || (pg == (int)N - 1 && leaf[t.first] == 1 && po < pg);
              }
              if (bad) {
                if (dbg && getenv("ZKWG_DEBUG_CHAIN_RULES") && gates[g].slot < net.names.size() && !(t.first & SRC_INPUT) && gates[t.first].slot < net.names.size())
                  fprintf(stderr, "[zkwg] %s:   %s  reads  %s\n", tag, net.names[gates[g].slot].c_str(), net.names[gates[t.first].slot].c_str());
                if (dbg && getenv("ZKWG_DEBUG_CHAIN_RULES"))
                  fprintf(stderr, "[zkwg] %s: gate %u (op %u, pos %d, line %d) leaves: operand %s%u (pos %d, cand %d, leaf %d, chain %d)\n", tag, g, gates[g].op, pos[g],
                          gates[g].at ? gates[g].at->line : -1, (t.first & SRC_INPUT) ? "byte " : "gate ", t.first & 0x1fffffffu,
                          (t.first & SRC_INPUT) ? tp((int)(t.first & 0x1fffffffu)) : pos[t.first], (t.first & SRC_INPUT) ? 0 : (int)cand[t.first],
                          (t.first & SRC_INPUT) ? 0 : (int)leaf[t.first], (t.first & SRC_INPUT) ? 0 : (int)g_chain[t.first]);
                cand[g] = 0; changed = true; break;
              }
            }
        }
      }
      std::vector<u32> cnt(N, 0), nz;
      for (u32 g = 0; g < n; ++g) if (cand[g]) ++cnt[pos[g]];
      for (u32 c : cnt) if (c) nz.push_back(c);
      if (nz.size() < 8) {
        if (dbg) { u64 nc = 0; for (u32 g = 0; g < n; ++g) nc += cand[g]; fprintf(stderr, "[zkwg] %s: %llu candidates at %zu positions\n", tag, (unsigned long long)nc, nz.size()); }
        return give_up("no per-position blocks");
      }
      std::nth_element(nz.begin(), nz.begin() + nz.size() / 2, nz.end());
      const u32 med = nz[nz.size() / 2];
      u32 cut = end;
      for (u32 i = bwd ? 2u : 0u; i < end; ++i) if (cnt[i] > 4 * med + 16) { cut = i; break; }   // (the mirrored walk starts with the block the forward pass left)
      if (cut == end) break;
      end = cut;
      for (u32 g = 0; g < n; ++g) if (cand[g] && pos[g] >= (int)end) cand[g] = 0;
    }
    if (end < 8) return give_up("the blocks are not bounded");
    // what a chain gate reads, and until which position a value of an earlier position is read (carried)
    std::vector<int> last(n, -1);
    for (u32 g = 0; g < n; ++g) {
      if (!cand[g]) continue;
      for (int i = 0; i < nforms(gates[g]); ++i)
        for (auto& t : gates[g].f[i].t) {
          if (t.first & SRC_INPUT) continue;
          const u32 o = t.first;
          if (sup[o] == -1) continue;
          if (pos[o] < pos[g]) last[o] = std::max(last[o], pos[g]);
        }
    }
    // chain gates the list reads: mask bits
    std::vector<u8> cf(n, 0);
    for (u32 g = 0; g < n; ++g) {
      if (!in_list(g)) continue;
      for (int i = 0; i < nforms(gates[g]); ++i)
        for (auto& t : gates[g].f[i].t)
          if (!(t.first & SRC_INPUT) && cand[t.first]) cf[t.first] = 1;
    }
    std::vector<std::vector<u32>> blk(end), carry(end);
    u32 n_cand = 0;
    for (u32 g = 0; g < n; ++g) {
      if (pos[g] < 0 || pos[g] >= (int)end) continue;
      if (cand[g]) { blk[pos[g]].push_back(g); ++n_cand; }
      if ((cand[g] || leaf[g]) && last[g] > pos[g]) carry[pos[g]].push_back(g);
    }
    if (!n_cand) return give_up("nothing to collapse");
    // operand roles inside a position
    enum : i64 { R_CONST = 0, R_STATE = 1, R_BLOCK = 2, R_LOCAL = 3, R_BYTE = 4, R_FWD = 5 };
    // A class = one block descriptor.  Its tables are rows indexed by the state, and the state is the index of the carried
    // valuation in ONE dictionary for the whole message (discovery order; the empty valuation entering the first position is
    // state 0), so a class is valid at every position with its descriptor whatever the set of states reachable there; rows are
    // tabulated when a position first reaches them.  A row has fdim x 256 entries: one per symbol (forward state, byte).
    struct Class {
      std::vector<i64> desc;
      std::vector<u8> have;                     // [state]: row tabulated
      std::vector<std::vector<u8>> delta;       // [state][symbol]
      std::vector<std::vector<u32>> words;      // [state][block gate * symbols + symbol]
      std::vector<std::vector<u8>> valid;       // [state][symbol]: the symbol can follow the state (backward pass; forward: all)
    };
    const u32 NSYM = fdim * 256;
    std::vector<Class> classes;
    std::map<std::vector<i64>, u32> id_of;      // valuation -> state
    std::vector<std::vector<i64>> val_of;       // state -> valuation
    // (backward pass: a valuation ends with the forward state entering the NEXT byte, -1 = any: the symbol (f, b) of a byte can only
    // precede it if the forward chain steps from f to that state on b -- without this the tables would be enumerated over
    // combinations that never occur, and values that are boolean on every real message would not look it)
    { std::vector<i64> v0; if (bwd) v0.push_back(-1); id_of.emplace(v0, 0u); val_of.push_back(v0); }
    std::vector<u8> class_at(N, 0);
    std::vector<u32> reach_bits((size_t)N * 8, 0);
    std::vector<u32> reach(1, 0u), prev_reach;  // states entering the position (sorted)
    std::vector<u32> S;                         // carried gates entering the position, in order
    std::vector<u32> cls_of(end, 0);
    std::vector<i64> role(n, -1), ridx(n, 0);   // scratch: role of a gate at the current position
    u32 max_bits = 0;
    for (u32 i = 0; i < end; ++i) {
      const u32 at = (u32)tp((int)i);           // the message byte of this position
      const std::vector<u32>& B = blk[i];
      std::vector<u32> L, touched;
      auto set_role = [&](u32 g, i64 r, i64 ix) { role[g] = r; ridx[g] = ix; touched.push_back(g); };
      for (u32 k = 0; k < S.size(); ++k) set_role(S[k], R_STATE, k);
      for (u32 k = 0; k < B.size(); ++k) set_role(B[k], R_BLOCK, k);
      auto leaf_ref = [&](u32 g) { if (role[g] < 0) { set_role(g, leaf[g] == 2 ? R_FWD : R_LOCAL, (i64)L.size()); L.push_back(g); } };
      auto leaf_name = [&](u32 g) -> i64 { return leaf[g] == 2 ? ((i64)F.cls[at] << 32 | (i64)cbit_fwd[g]) : (i64)sig[g]; };
      // carried set leaving the position
      std::vector<u32> Sn;
      for (u32 g : S) if (last[g] > (int)i) Sn.push_back(g);
      for (u32 g : carry[i]) Sn.push_back(g);
      // descriptor
      std::vector<i64> D;
      D.push_back((i64)S.size()); D.push_back((i64)B.size()); D.push_back((i64)Sn.size());
      bool ok = true;
      u32 nbits = 0;
      auto name_operand = [&](u32 o) {
        if (role[o] < 0) { if (leaf[o] && pos[o] == (int)i) leaf_ref(o); else { ok = false; return; } }
        D.push_back(role[o]); D.push_back(role[o] == R_LOCAL || role[o] == R_FWD ? leaf_name(o) : ridx[o]);
      };
      for (u32 g : B) {
        const Gate& G = gates[g];
        D.push_back((i64)G.op); D.push_back(G.k); D.push_back((G.slot < net.n_kept ? 1 : 0) | (cf[g] ? 2 : 0));
        if (cf[g]) ++nbits;
        for (int f = 0; f < nforms(G); ++f) {
          D.push_back(G.f[f].c0); D.push_back((i64)G.f[f].t.size());
          for (auto& t : G.f[f].t) {
            D.push_back(t.second);
            if (t.first & SRC_INPUT) { D.push_back(R_BYTE); D.push_back(0); continue; }
            if (sup[t.first] == -1) { ++now; u32 w; D.push_back(R_CONST); D.push_back(eval(t.first, 0, w)); continue; }
            name_operand(t.first);
          }
        }
      }
      for (u32 g : Sn) name_operand(g);
      if (!ok) return give_up("internal: a chain operand without a role");
      // (backward pass) which symbols (forward state, byte) can occur at a position depends on the position: whether the forward chain
      // covers it, the forward class there (its transition function prunes the enumeration) -- part of the class identity; and the
      // forward states that can enter it, which vary from position to position: cells are tabulated as positions need them (below)
      if (bwd) { D.push_back(at >= F.end ? -1 : (i64)F.cls[at]); }
      max_bits = std::max(max_bits, nbits);
      int use = -1;
      if (i > 0 && classes[cls_of[i - 1]].desc == D) use = (int)cls_of[i - 1];
      for (size_t c = 0; use < 0 && c < classes.size(); ++c) if (classes[c].desc == D) use = (int)c;
      if (use < 0) {
        if (classes.size() >= 255) return give_up("more than 255 position classes");
        use = (int)classes.size();
        classes.emplace_back();
        classes.back().desc = D;
        if (dbg && classes.size() <= 4)
          fprintf(stderr, "[zkwg] %s: byte %u opens class %d: %zu carried in, %zu gates, %zu carried out, %zu leaves\n", tag, at, use, S.size(), B.size(), Sn.size(), L.size());
      }
      cls_of[i] = (u32)use;
      class_at[at] = (u8)use;
      for (u32 st : reach) reach_bits[(size_t)at * 8 + st / 32] |= 1u << (st % 32);
      // can symbol y follow state st at this position?  (forward pass: every byte)
      auto needed = [&](u32 st, u32 y) -> bool {
        if (!bwd) return true;
        const u32 f = y / 256, b = y % 256;
        if (at >= F.end) return f == 0;
        if (!((F.reach_bits[(size_t)at * 8 + f / 32] >> (f % 32)) & 1u)) return false;
        const i64 nxt = val_of[st].back();
        return nxt < 0 || (i64)F.delta[((size_t)F.cls[at] * F.smax + f) * 256 + b] == nxt;
      };
      bool same_fwd = true;     // (backward pass) the same forward states enter this byte as the previous one
      if (bwd && i > 0) {
        const u32 pat = (u32)tp((int)i - 1);
        same_fwd = (at >= F.end) == (pat >= F.end);
        for (u32 k = 0; k < 8 && same_fwd && at < F.end; ++k) same_fwd = F.reach_bits[(size_t)at * 8 + k] == F.reach_bits[(size_t)pat * 8 + k];
      }
      const bool same = i > 0 && cls_of[i - 1] == (u32)use && reach == prev_reach && same_fwd;   // same function, same states, same symbols: same successors
      if (!same) {
        Class& C = classes[use];
        // cells (state, symbol) this position can meet and no earlier position of the class tabulated: a row is NOT complete once some
        // position reached its state -- the symbols that can follow a state differ between positions (the forward states entering
        // them do: anchored or periodic automata, the first and last bytes)
        std::vector<u32> todo;
        for (u32 st : reach) {
          if (C.have.size() <= st) { C.have.resize(st + 1, 0); C.delta.resize(st + 1); C.words.resize(st + 1); C.valid.resize(st + 1); }
          if (!C.have[st]) { C.have[st] = 1; C.delta[st].assign(NSYM, 0); C.words[st].assign(B.size() * NSYM, 0); C.valid[st].assign(NSYM, 0); }
          bool miss = false;
          for (u32 y = 0; y < NSYM && !miss; ++y) miss = !C.valid[st][y] && needed(st, y);
          if (miss) todo.push_back(st);
        }
        if (!todo.empty()) {
          // leaves per symbol: byte-local gates by the byte, forward-chain bits by (forward state, byte)
          std::vector<std::vector<long long>> LV(L.size());
          for (size_t l = 0; l < L.size(); ++l) LV[l].assign(leaf[L[l]] == 2 ? NSYM : 256, 0);
          for (u32 b = 0; b < 256; ++b) { ++now; for (size_t l = 0; l < L.size(); ++l) if (leaf[L[l]] != 2) { u32 w; LV[l][b] = eval(L[l], (long long)b, w); } }
          for (size_t l = 0; l < L.size(); ++l)
            if (leaf[L[l]] == 2) {
              const u32 bit = (u32)cbit_fwd[L[l]];
              for (u32 y = 0; y < NSYM; ++y)
                LV[l][y] = (F.mask[(((size_t)F.cls[at] * F.smax + y / 256) * 256 + y % 256) * F.mask_words + bit / ZKC_MASK_BITS] >> (bit % ZKC_MASK_BITS)) & 1u;
            }
          std::map<u32, long long> consts;
          for (u32 g : B)
            for (int q = 0; q < nforms(gates[g]); ++q)
              for (auto& t : gates[g].f[q].t)
                if (!(t.first & SRC_INPUT) && sup[t.first] == -1 && !consts.count(t.first)) { ++now; u32 w; consts[t.first] = eval(t.first, 0, w); }
          std::vector<long long> bv(B.size());
          std::vector<i64> nv;
          for (u32 st : todo) {
            const std::vector<i64> sv = val_of[st];
            if (sv.size() != S.size() + (bwd ? 1 : 0)) return give_up("internal: a state of another shape reaches the position");
            for (u32 y = 0; y < NSYM; ++y) {
              const u32 b = y % 256;
              if (C.valid[st][y] || !needed(st, y)) continue;
              C.valid[st][y] = 1;
              auto value_of = [&](u32 o) -> long long {
                switch (role[o]) { case R_STATE: return sv[ridx[o]]; case R_BLOCK: return bv[ridx[o]]; case R_FWD: return LV[ridx[o]][y]; default: return LV[ridx[o]][b]; }
              };
              for (u32 k = 0; k < B.size(); ++k) {
                const Gate& G = gates[B[k]];
                long long f[3] = {0, 0, 0};
                for (int q = 0; q < nforms(G); ++q) {
                  f[q] = G.f[q].c0;
                  for (auto& t : G.f[q].t)
                    f[q] += t.second * ((t.first & SRC_INPUT) ? (long long)b : (sup[t.first] == -1 ? consts[t.first] : value_of(t.first)));
                }
                long long r = f[0];
                if (G.op == G_QUAD) r = f[0] * f[1] + f[2];
                else if (G.op == G_NEZ) r = (f[0] != 0 ? G.k : 0) + f[1];
                else if (G.op == G_BIT) r = f[0] < 0 ? -1 : ((f[0] >> G.k) & 1);
                u32 word;
                if (G.op == G_INV0) {
                  if (std::llabs(f[0]) > (long long)net.inv_need) return give_up("an inverse hint of the chain is outside the table");
                  word = f[0] == 0 ? 0u : (VAL_INVERSE | ((u32)f[0] & 0x7fffffffu));
                } else {
                  if (r <= -(1ll << 30) || r >= (1ll << 30)) return give_up("a chain value is outside the stored range");
                  word = (u32)r & 0x7fffffffu;
                }
                bv[k] = r;
                C.words[st][(size_t)k * NSYM + y] = word;
              }
              nv.clear();
              for (u32 g : Sn) nv.push_back((i64)value_of(g));
              if (bwd) nv.push_back(at < F.end ? (i64)(y / 256) : -1);
              auto it = id_of.find(nv);
              if (it == id_of.end()) {
                if (val_of.size() >= 255) return give_up("more than 255 chain states");
                it = id_of.emplace(nv, (u32)val_of.size()).first;
                val_of.push_back(nv);
              }
              C.delta[st][y] = (u8)it->second;
            }
          }
        }
        // successors over the symbols that can occur here
        std::vector<u8> seen(256, 0);
        for (u32 st : reach) for (u32 y = 0; y < NSYM; ++y) if (C.valid[st][y] && needed(st, y)) seen[C.delta[st][y]] = 1;
        prev_reach.swap(reach);
        reach.clear();
        for (u32 q = 0; q < 256; ++q) if (seen[q]) reach.push_back(q);
      }
      S.swap(Sn);
      for (u32 g : touched) role[g] = -1;
    }
    // the booleans taken on trust
    if (trust) {
      bool grew = false;
      for (u32 i = 0; i < end; ++i) {
        const Class& C = classes[cls_of[i]];
        for (u32 k = 0; k < blk[i].size(); ++k) {
          if (!cf[blk[i][k]]) continue;
          bool bad = false;
          for (u32 st = 0; st < C.have.size() && !bad; ++st)
            if (C.have[st]) for (u32 y = 0; y < NSYM; ++y) if (C.valid[st][y] && C.words[st][(size_t)k * NSYM + y] > 1u) { bad = true; break; }
          if (bad) { nonbool[blk[i][k]] = 1; grew = true; }
        }
      }
      if (grew) { if (dbg) fprintf(stderr, "[zkwg] %s: a value the list reads is not boolean; again without it\n", tag); return 2; }
      for (u32 i = 0; i < end; ++i) for (u32 g : blk[i]) if (cf[g]) { gates[g].lo = 0; gates[g].hi = 1; }   // (verified over every reachable state)
    }
    // tables, indexed by the CORE state: the carried values without the forward state that only pruned the enumeration (what a
    // block computes depends on the values and the symbol alone, so rows of states with the same core agree wherever both are valid)
    std::vector<u32> core_of(val_of.size());
    {
      std::map<std::vector<i64>, u32> core_id;
      for (size_t st = 0; st < val_of.size(); ++st) {
        std::vector<i64> v = val_of[st];
        if (bwd) v.pop_back();
        core_of[st] = core_id.emplace(v, (u32)core_id.size()).first->second;
      }
    }
    u32 smax = *std::max_element(core_of.begin(), core_of.end()) + 1;
    const u32 mw = (max_bits + ZKC_MASK_BITS - 1) / ZKC_MASK_BITS;
    // Classes -> groups.  Two descriptors can differ and still describe the same functions (the first positions of a message cut
    // their long sums into partial sums differently: other temporaries, same signals); their cells -- (core state, symbol) ->
    // next state, the kept gates' words in order, the list-read booleans in order -- then agree wherever both were tabulated, and
    // one set of tables serves both (a cell only one of them reached is taken from that one: the other never looks it up).
    struct Cells {
      u32 nkept = 0, ncf = 0;
      std::vector<u8> set, delta, cb;   // [core][symbol]; cb: [cf gate][core][symbol]
      std::vector<u32> kw;              // [kept gate][core][symbol]
    };
    size_t CELLS = (size_t)smax * NSYM;
    std::vector<int> rep(classes.size(), -1);
    for (u32 i = 0; i < end; ++i) if (rep[cls_of[i]] < 0) rep[cls_of[i]] = (int)i;
    std::vector<Cells> groups;
    std::vector<u32> group_of(classes.size(), 0);
    for (size_t c = 0; c < classes.size(); ++c) {
      const Class& C = classes[c];
      const std::vector<u32>& Bk = blk[rep[c]];
      Cells X;
      for (u32 g : Bk) { if (gates[g].slot < net.n_kept) ++X.nkept; if (cf[g]) ++X.ncf; }
      X.set.assign(CELLS, 0); X.delta.assign(CELLS, 0); X.kw.assign(X.nkept * CELLS, 0); X.cb.assign(X.ncf * CELLS, 0);
      for (u32 st = 0; st < C.have.size(); ++st) {
        if (!C.have[st]) continue;
        for (u32 y = 0; y < NSYM; ++y) {
          if (!C.valid[st][y]) continue;
          const size_t cell = (size_t)core_of[st] * NSYM + y;
          X.set[cell] = 1; X.delta[cell] = (u8)core_of[C.delta[st][y]];
          u32 jk = 0, jq = 0;
          for (u32 k = 0; k < Bk.size(); ++k) {
            const u32 w = C.words[st][(size_t)k * NSYM + y];
            if (gates[Bk[k]].slot < net.n_kept) X.kw[jk++ * CELLS + cell] = w;
            if (cf[Bk[k]]) X.cb[jq++ * CELLS + cell] = (u8)(w & 1u);
          }
        }
      }
      int into = -1;
      for (size_t gi = 0; gi < groups.size() && into < 0; ++gi) {
        const Cells& G = groups[gi];
        if (G.nkept != X.nkept || G.ncf != X.ncf) continue;
        bool same = true;
        for (size_t cell = 0; cell < CELLS && same; ++cell) {
          if (!G.set[cell] || !X.set[cell]) continue;
          same = G.delta[cell] == X.delta[cell];
          for (u32 q = 0; q < X.nkept && same; ++q) same = G.kw[q * CELLS + cell] == X.kw[q * CELLS + cell];
          for (u32 q = 0; q < X.ncf && same; ++q) same = G.cb[q * CELLS + cell] == X.cb[q * CELLS + cell];
        }
        if (same) into = (int)gi;
      }
      if (into < 0) { group_of[c] = (u32)groups.size(); groups.push_back(std::move(X)); continue; }
      Cells& G = groups[into];
      for (size_t cell = 0; cell < CELLS; ++cell) {
        if (G.set[cell] || !X.set[cell]) continue;
        G.set[cell] = 1; G.delta[cell] = X.delta[cell];
        for (u32 q = 0; q < X.nkept; ++q) G.kw[q * CELLS + cell] = X.kw[q * CELLS + cell];
        for (u32 q = 0; q < X.ncf; ++q) G.cb[q * CELLS + cell] = X.cb[q * CELLS + cell];
      }
      group_of[c] = (u32)into;
    }
    // States that no table can tell apart are one state (Moore's partition refinement over all classes: same tabulated cells, same
    // words and bits in them, successors in the same block).  The carried set often holds more than the automaton needs -- an
    // intermediate of the next state beside the state itself -- and every valuation of it was numbered: 132 states for the 4-state
    // automaton of tests/golden/regex_style/simple_regex.circom.  Fewer states = fewer rows here and fewer symbols per row in the
    // backward pass.
    {
      std::vector<u32> block(smax, 0), next(smax, 0);
      u32 n_blocks = 1;
      for (int round = 0; round < 300; ++round) {
        std::map<std::vector<u32>, u32> ids;
        for (u32 st = 0; st < smax; ++st) {
          std::vector<u32> key;
          key.push_back(block[st]);
          for (const Cells& G : groups)
            for (u32 y = 0; y < NSYM; ++y) {
              const size_t cell = (size_t)st * NSYM + y;
              if (!G.set[cell]) { key.push_back(0xffffffffu); continue; }
              key.push_back(block[G.delta[cell]]);
              if (round == 0) {       // (the contents separate the states once; afterwards only the successors' blocks change)
                for (u32 q = 0; q < G.nkept; ++q) key.push_back(G.kw[q * CELLS + cell]);
                for (u32 q = 0; q < G.ncf; ++q) key.push_back(G.cb[q * CELLS + cell]);
              }
            }
          next[st] = ids.emplace(std::move(key), (u32)ids.size()).first->second;
        }
        const bool stable = ids.size() == n_blocks && round > 0;
        n_blocks = (u32)ids.size();
        block.swap(next);
        if (stable) break;
      }
      if (n_blocks < smax) {
        // (blocks are numbered in state order, so state 0 -- the start -- stays 0)
        const size_t NC = (size_t)n_blocks * NSYM;
        for (Cells& G : groups) {
          Cells H;
          H.nkept = G.nkept; H.ncf = G.ncf;
          H.set.assign(NC, 0); H.delta.assign(NC, 0); H.kw.assign(G.nkept * NC, 0); H.cb.assign(G.ncf * NC, 0);
          for (u32 st = 0; st < smax; ++st)
            for (u32 y = 0; y < NSYM; ++y) {
              const size_t a = (size_t)st * NSYM + y, b = (size_t)block[st] * NSYM + y;
              if (!G.set[a]) continue;
              H.set[b] = 1; H.delta[b] = (u8)block[G.delta[a]];
              for (u32 q = 0; q < G.nkept; ++q) H.kw[q * NC + b] = G.kw[q * CELLS + a];
              for (u32 q = 0; q < G.ncf; ++q) H.cb[q * NC + b] = G.cb[q * CELLS + a];
            }
          G = std::move(H);
        }
        std::vector<u32> rb((size_t)N * 8, 0);
        for (u32 at = 0; at < N; ++at)
          for (u32 st = 0; st < smax && st < 256; ++st)
            if ((reach_bits[(size_t)at * 8 + st / 32] >> (st % 32)) & 1u) rb[(size_t)at * 8 + block[st] / 32] |= 1u << (block[st] % 32);
        if (!bwd) reach_bits.swap(rb);          // (recorded by precise state = core state in the forward pass; the backward pass reads them)
        if (dbg) fprintf(stderr, "[zkwg] %s: %u states are %u\n", tag, smax, n_blocks);
        smax = n_blocks; CELLS = NC;
      }
    }
    {
      // the tables are looked up once per slot and email: they have to stay cache-sized
      size_t words = 0;
      for (const Cells& G : groups) words += (size_t)G.nkept + 1 + mw;
      if (words * CELLS * 4 > (64u << 20)) {
        // not a silent fallback: the gate list would evaluate this recurrence correctly but an order of magnitude slower, and a
        // deployment should know (ZKWG_NET_ALLOW_LIST_FALLBACK=1 accepts it)
        char msg[200];
        snprintf(msg, sizeof msg, "the %s tables of the regex template would take %.0f MB (limit 64 MB: %u states x %u symbols x %zu words per cell); "
                 "set ZKWG_NET_ALLOW_LIST_FALLBACK=1 to evaluate the recurrence gate by gate instead", tag, words * CELLS * 4 / 1e6, smax, NSYM, words);
        if (!getenv("ZKWG_NET_ALLOW_LIST_FALLBACK") || !atoi(getenv("ZKWG_NET_ALLOW_LIST_FALLBACK"))) fail(msg);
        fprintf(stderr, "[zkwg] warning: %s\n", msg);
        return give_up("the tables would not stay in the cache");
      }
    }
    T.end = end; T.smax = smax; T.classes = (u32)groups.size(); T.mask_words = mw; T.fdim = fdim;
    T.reach_bits = reach_bits;
    T.cls.assign(N, 0);
    T.delta.assign(groups.size() * CELLS, 0);
    T.mask.assign(groups.size() * CELLS * mw, 0);
    std::map<std::vector<u32>, u32> tab_of;     // table content -> index
    std::vector<std::vector<u32>> fn_of(groups.size());   // [group][kept gate] -> table
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      const Cells& G = groups[gi];
      std::copy(G.delta.begin(), G.delta.end(), T.delta.begin() + gi * CELLS);
      for (u32 q = 0; q < G.ncf; ++q)
        for (size_t cell = 0; cell < CELLS; ++cell)
          if (G.cb[q * CELLS + cell]) T.mask[(gi * CELLS + cell) * mw + q / ZKC_MASK_BITS] |= 1u << (q % ZKC_MASK_BITS);
      for (u32 q = 0; q < G.nkept; ++q) {
        std::vector<u32> w(G.kw.begin() + q * CELLS, G.kw.begin() + (q + 1) * CELLS);
        auto it = tab_of.find(w);
        if (it == tab_of.end()) {
          it = tab_of.emplace(w, (u32)(T.tab.size() / CELLS)).first;
          T.tab.insert(T.tab.end(), w.begin(), w.end());
        }
        if (it->second >= 0x1fffu) fail("too many distinct chain tables in the regex template");
        fn_of[gi].push_back(it->second);
      }
    }
    const size_t tab_words = CELLS;
    for (u32 i = 0; i < end; ++i) {
      const u32 gi = group_of[cls_of[i]], at = (u32)tp((int)i);
      if (at >= 0x10000) fail("the regex template's message is too long for the chain tables");
      T.cls[at] = (u8)gi;
      u32 jk = 0, jq = 0;
      for (u32 g : blk[i]) {
        g_chain[g] = (u8)(1 + pass); ++T.n_gates;
        if (cf[g]) { cbit[g] = (int)jq++; g_fpos[g] = (int)at; ++T.n_front; }     // bit of the position's chain mask words
        // (a gate the list reads keeps a BIT record there, which writes the same word; it gets its table all the same, so that it can
        // leave the list when a later pass takes its readers)
        if (gates[g].slot < net.n_kept) net.slot_desc[gates[g].slot] = (bwd ? 0xE0000000u : 0xC0000000u) | (fn_of[gi][jk++] << 16) | at;
      }
    }
    if (dbg)
      fprintf(stderr, "[zkwg] %s: %u positions collapsed: %u gates (%u read by the list through %u mask word(s)), %zu descriptors in %zu classes, <= %u states x %u symbols, %zu tables (%.1f MB)\n",
              tag, end, T.n_gates, T.n_front, mw, classes.size(), groups.size(), smax, NSYM, T.tab.size() / tab_words, T.tab.size() * 4 / 1e6);
    return 1;
  }

  // Evaluation schedule.  Gates are list-scheduled into chunks of mutually independent gates inside a sliding
  // window (so the comparators of the next message byte fill the chunks of the current byte's state recurrence);
  // a chunk is executed in steps of up to 64 gates, one per lane.  Where an operand lives is decided here, by
  // simulating the evaluator's direct-mapped value cache: the cache (recent values) or a pinned region for values
  // that are read long after they were produced (backward chains over the whole message).
  void emit(Net& net, u32 force_lanes = 0) {
    std::vector<u32> chunk_of(gates.size(), 0);
    std::vector<std::vector<u32>> steps;
    net.n_gates = (u32)gates.size();
    auto nforms = [](const Gate& g) { return (g.op == G_QUAD || g.op == G_ASSERT) ? 3 : (g.op == G_NEZ ? 2 : 1); };
    // list scheduling straight into steps of <= 64 gates: a gate goes to the earliest open step after all its operands
    // that still has a free lane; `window` + 1 steps stay open, older ones are closed in order
    const u32 window = 12;
    // gates per step = lanes per email of zk_net_eval: with the byte-local gates gone (localize) a step holds ~15 gates, and
    // 32 lanes per step cost +2 % steps (16: +27 %) -- two emails share a wavefront
    // -- and once the state recurrence is served from tables (chain_pass) what is left are a few gates per position:
    // 16 lanes, four emails per wavefront, when four LDS images fit
    const u32 mask_stride = net.mask_words + net.chain.mask_words + net.bchain.mask_words;
    const u32 lanes_default = (net.chain.end && 16ull * (n_in * (1 + mask_stride) + 4096) < 150u * 1024u) ? 16u : 32u;
    u32 step_lanes = getenv("ZKWG_NET_LANES") ? (u32)atoi(getenv("ZKWG_NET_LANES")) : lanes_default;
    if (step_lanes != 16 && step_lanes != 32 && step_lanes != 64) step_lanes = lanes_default;
    if (force_lanes) step_lanes = force_lanes;
    net.lanes = step_lanes;
    std::deque<std::vector<u32>> open;
    u32 base = 1;
    std::vector<u32> outs;
    for (u32 gi = 0; gi < gates.size(); ++gi) {
      const Gate& g = gates[gi];
      if (g.op == G_OUT) { outs.push_back(gi); continue; }   // the outputs go last, 64 per step (they take the 64-bit path)
      if (g_skip[gi]) continue;                              // byte-local: not evaluated per email (localize)
      if (g_chain[gi] && g_front[gi] < 0) continue;          // a function of (chain state, symbol): served from the chain tables (chain_pass)
      u32 c = base;
      if (g_front[gi] < 0)                                   // (a frontier gate only reads its byte's mask word)
        for (int i = 0; i < nforms(g); ++i)
          for (auto& t : g.f[i].t) if (!(t.first & SRC_INPUT)) c = std::max(c, chunk_of[t.first] + 1);
      for (;;) {
        while (c >= base + open.size()) open.emplace_back();
        if (open[c - base].size() < step_lanes) break;
        ++c;
      }
      chunk_of[gi] = c;
      open[c - base].push_back(gi);
      while (open.size() > window + 1) { steps.push_back(std::move(open.front())); open.pop_front(); ++base; }
    }
    while (!open.empty()) { steps.push_back(std::move(open.front())); open.pop_front(); }
    for (size_t b = 0; b < outs.size(); b += step_lanes) steps.emplace_back(outs.begin() + b, outs.begin() + std::min(outs.size(), b + step_lanes));
    {
      std::vector<std::vector<u32>> nonempty;
      for (auto& st : steps) if (!st.empty()) nonempty.push_back(std::move(st));
      steps.swap(nonempty);
    }
    net.n_chunks = (u32)steps.size();
    // LDS words by liveness: a value gets a word when it is produced and gives it back after its last reader
    // (a step reads before it writes, so the word of a value last read in step t can be rewritten in step t).
    // Values nobody reads (most kept signals: they only go to the witness) get no word at all.
    std::vector<u32> step_of(gates.size(), 0), last_use(gates.size(), 0);
    for (u32 t = 0; t < steps.size(); ++t) for (u32 gi : steps[t]) step_of[gi] = t;
    std::vector<u8> used(gates.size(), 0);
    for (u32 t = 0; t < steps.size(); ++t)
      for (u32 gi : steps[t]) {
        const Gate& g = gates[gi];
        if (g_front[gi] >= 0) continue;
        for (int i = 0; i < nforms(g); ++i)
          for (auto& tm : g.f[i].t) if (!(tm.first & SRC_INPUT)) { used[tm.first] = 1; last_use[tm.first] = std::max(last_use[tm.first], t); }
      }
    std::vector<u32> word_of(gates.size(), 0xffffffffu), free_words;
    std::vector<std::pair<u32, u32>> live;   // min-heap on (last use, word)
    auto cmp = [](const std::pair<u32, u32>& x, const std::pair<u32, u32>& y) { return x.first > y.first; };
    u32 hwm = 0;
    for (u32 t = 0; t < steps.size(); ++t) {
      while (!live.empty() && live.front().first <= t) { free_words.push_back(live.front().second); std::pop_heap(live.begin(), live.end(), cmp); live.pop_back(); }
      for (u32 gi : steps[t]) {
        if (!used[gi]) continue;
        u32 w;
        if (!free_words.empty()) { w = free_words.back(); free_words.pop_back(); } else w = hwm++;
        word_of[gi] = w;
        live.emplace_back(last_use[gi], w);
        std::push_heap(live.begin(), live.end(), cmp);
      }
    }
    // LDS words: [values | message bytes | 0 | scratch | masks: mask_words per message byte]
    const u32 lds_msg = hwm, lds_zero = lds_msg + n_in, lds_dummy = lds_zero + 1;
    net.n_pins = hwm;
    net.lds_masks = lds_dummy + 1;
    net.lds_words = net.lds_masks + n_in * mask_stride;
    if (4ull * net.lds_words * (64u / net.lanes) + 16u > 150u * 1024u && net.lanes == 16 && !getenv("ZKWG_NET_LANES")) return emit(net, 32);
    if (4ull * net.lds_words * (64u / net.lanes) + 16u > 150u * 1024u)
      fail("the evaluator's LDS image (" + std::to_string(net.lds_words) + " words per email, " + std::to_string(64u / net.lanes) + " emails per wavefront) would exceed the 160 KB of a gfx950 CU: set ZKWG_NET_LANES=64");
    if (net.lds_words > 16000) fail("the template keeps " + std::to_string(hwm) + " values alive at once; the evaluator's LDS image would exceed 64 KiB");
    // records
    net.records.clear(); net.step_count.clear();
    auto span = [&](u32 src, i64& lo, i64& hi) { if (src & SRC_INPUT) { lo = in_lo[src & 0x1fffffffu]; hi = in_hi[src & 0x1fffffffu]; } else { lo = gates[src].lo; hi = gates[src].hi; } };
    for (auto& st : steps) {
      u32 general = 0, half = 0x4000;
      for (u32 gi : st) {
        if (g_front[gi] >= 0) {
          // frontier gate: bit (index % ZKC_MASK_BITS) of mask word (index / ZKC_MASK_BITS) of its byte, as a BIT record
          const Gate& g = gates[gi];
          const long long byte = g_fpos[gi];
          if (byte < 0) fail("internal: frontier gate without a message byte");
          const u32 b = (u32)g_front[gi], wordi = net.lds_masks + (u32)byte * mask_stride + b / ZKC_MASK_BITS;
          u32 r[16] = {0};
          r[0] = G_BIT | ((b % ZKC_MASK_BITS) << 4);
          r[1] = g.slot;
          r[3] = word_of[gi] == 0xffffffffu ? lds_dummy : word_of[gi];
          for (u32 q = 0; q < 8; ++q) r[8 + q] = (lds_zero << 2);
          r[8] = (wordi << 2) | (1u << 18);
          net.records.insert(net.records.end(), r, r + 16);
          continue;
        }
        const Gate& g = gates[gi];
        const int nf = nforms(g);
        bool wide = false;
        for (int i = 0; i < nf; ++i) for (auto& t : g.f[i].t) wide = wide || t.second < -8191 || t.second > 8191;
        // the 32-bit path of the evaluator is exact when the value intervals prove it: operands and results below
        // 2^20 in magnitude, every sum of products below 2^30; everything else takes the 64-bit path
        bool exact32 = !wide && g.op != G_ASSERT && g.op != G_OUT;
        const i64 B20 = (i64)1 << 20;
        __int128 mag[3] = {0, 0, 0};
        for (int i = 0; i < nf && exact32; ++i) {
          mag[i] = g.f[i].c0 < 0 ? -(__int128)g.f[i].c0 : g.f[i].c0;
          for (auto& t : g.f[i].t) {
            i64 lo, hi;
            span(t.first, lo, hi);
            const i64 m = std::max<i64>(std::llabs(lo), std::llabs(hi));
            if (m >= B20) exact32 = false;
            mag[i] += (__int128)(t.second < 0 ? -t.second : t.second) * m;
          }
          if (mag[i] >= ((__int128)1 << 30)) exact32 = false;
        }
        if (exact32 && g.op == G_QUAD && (mag[0] * mag[1] + mag[2] >= ((__int128)1 << 30) || mag[0] >= (1 << 23) || mag[1] >= (1 << 23))) exact32 = false;
        if (exact32 && g.op == G_NEZ && (std::llabs(g.k) >= B20)) exact32 = false;
        if (exact32 && g.op == G_BIT) { i64 lo, hi; interval_of(g.f[0], lo, hi); if (lo < 0) exact32 = false; }
        if (exact32 && g.op == G_INV0 && std::max<i64>(std::llabs(g.lo), std::llabs(g.hi)) > (i64)net.inv_need) exact32 = false;
        if (!exact32) { general = 0x8000; ++net.n_general; }
        auto where = [&](u32 src) -> u32 {
          if (src & SRC_INPUT) return lds_msg + (src & 0x1fffffffu);
          if (word_of[src] == 0xffffffffu) fail("internal: operand without an LDS word");
          return word_of[src];
        };
        u32 r[16] = {0};
        r[0] = g.op | ((u32)(g.op == G_BIT ? g.k : 0) << 4) | (wide ? 1u << 9 : 0u);
        r[1] = g.op == G_OUT ? g.out_index : (g.op == G_ASSERT ? 0u : g.slot);
        r[2] = (u32)(int32_t)(g.op == G_NEZ ? g.k : 0);
        r[3] = (g.op == G_OUT || g.op == G_ASSERT || g.op == G_INV0 || word_of[gi] == 0xffffffffu) ? lds_dummy : word_of[gi];
        if (g.op == G_ASSERT) ++net.n_asserts;
        // term slots: narrow 0,1 = A  2,3 = B  4..7 = C;  wide 0 = A  1 = B  2,3 = C;  a single sum fills them in order
        std::vector<std::pair<u32, i64>> slots(wide ? 4 : 8, std::make_pair(lds_zero, (i64)0));
        auto place = [&](const Lin& f, u32 first, u32 count) {
          if (f.t.size() > count) fail("internal: operand limit exceeded");
          for (size_t q = 0; q < f.t.size(); ++q) { slots[first + q] = std::make_pair(where(f.t[q].first), f.t[q].second); ++net.lds_hits; }
        };
        if (g.op == G_QUAD || g.op == G_ASSERT) {
          place(g.f[0], 0, wide ? 1 : 2); place(g.f[1], wide ? 1 : 2, wide ? 1 : 2); place(g.f[2], wide ? 2 : 4, wide ? 2 : 4);
          r[4] = (u32)(int32_t)g.f[0].c0; r[5] = (u32)(int32_t)g.f[1].c0; r[6] = (u32)(int32_t)g.f[2].c0;
        } else {
          place(g.f[0], 0, wide ? 4 : 8);
          r[4] = (u32)(int32_t)g.f[0].c0;
          if (g.op == G_NEZ) r[6] = (u32)(int32_t)g.f[1].c0;
        }
        for (size_t q = 4; q < slots.size(); ++q) if (slots[q].second != 0) half = 0;
        for (size_t q = 0; q < slots.size(); ++q) {
          if (wide) { r[8 + 2 * q] = slots[q].first; r[9 + 2 * q] = (u32)(int32_t)slots[q].second; }
          else r[8 + q] = (slots[q].first << 2) | ((u32)(int32_t)slots[q].second << 18);   // LDS byte offset | coefficient
        }
        net.records.insert(net.records.end(), r, r + 16);
      }
      net.step_count.push_back((u32)st.size() | general | (general ? 0u : half));
    }
    net.n_steps = (u32)steps.size();
    if (getenv("ZKWG_DEBUG_NET")) {
      u64 n_half = 0, n_full = 0, n_gen = 0, g_half = 0, g_full = 0;
      for (u32 c : net.step_count) { if (c & 0x8000) ++n_gen; else if (c & 0x4000) { ++n_half; g_half += c & 0x7f; } else { ++n_full; g_full += c & 0x7f; } }
      fprintf(stderr, "[zkwg] gate list: %llu steps with <= 4 terms (%llu gates), %llu with up to 8 (%llu gates), %llu on the 64-bit path\n",
              (unsigned long long)n_half, (unsigned long long)g_half, (unsigned long long)n_full, (unsigned long long)g_full, (unsigned long long)n_gen);
    }
    while (net.step_count.size() % 8) net.step_count.push_back(0);    // the evaluator works in groups of 8 steps
    for (int i = 0; i < 32; ++i) net.step_count.push_back(0);         // (and reads the counts up to three groups ahead)
    net.records.insert(net.records.end(), 64 * 16, 0u);   // the evaluator's lanes always load 64 records
    if (net.records.size() >= 0xffffffffull) fail("the gate list is too large");
  }
};

// circomlib 2.0.5 comparators / gates / bitify ([EXT], restated from their published definitions,
// SURVEY.md Appendix A.1); used only when the include path does not supply the files.
inline const char* Elab::builtin_source(const std::string& base) {
  if (base == "comparators.circom") return R"CIRCOM(
include "bitify.circom";
template IsZero() {
    signal input in;
    signal output out;
    signal inv;
    inv <-- in!=0 ? 1/in : 0;
    out <== -in*inv +1;
    in*out === 0;
}
template IsEqual() {
    signal input in[2];
    signal output out;
    component isz = IsZero();
    in[1] - in[0] ==> isz.in;
    isz.out ==> out;
}
template LessThan(n) {
    assert(n <= 252);
    signal input in[2];
    signal output out;
    component n2b = Num2Bits(n+1);
    n2b.in <== in[0]+ (1<<n) - in[1];
    out <== 1-n2b.out[n];
}
template LessEqThan(n) {
    signal input in[2];
    signal output out;
    component lt = LessThan(n);
    lt.in[0] <== in[0];
    lt.in[1] <== in[1]+1;
    lt.out ==> out;
}
template GreaterThan(n) {
    signal input in[2];
    signal output out;
    component lt = LessThan(n);
    lt.in[0] <== in[1];
    lt.in[1] <== in[0];
    lt.out ==> out;
}
template GreaterEqThan(n) {
    signal input in[2];
    signal output out;
    component lt = LessThan(n);
    lt.in[0] <== in[1];
    lt.in[1] <== in[0]+1;
    lt.out ==> out;
}
)CIRCOM";
  if (base == "bitify.circom") return R"CIRCOM(
template Num2Bits(n) {
    signal input in;
    signal output out[n];
    var lc1=0;
    var e2=1;
    for (var i = 0; i<n; i++) {
        out[i] <-- (in >> i) & 1;
        out[i] * (out[i] -1 ) === 0;
        lc1 += out[i] * e2;
        e2 = e2+e2;
    }
    lc1 === in;
}
)CIRCOM";
  if (base == "gates.circom") return R"CIRCOM(
template XOR() {
    signal input a;
    signal input b;
    signal output out;
    out <== a + b - 2*a*b;
}
template AND() {
    signal input a;
    signal input b;
    signal output out;
    out <== a*b;
}
template OR() {
    signal input a;
    signal input b;
    signal output out;
    out <== a + b - a*b;
}
template NOT() {
    signal input in;
    signal output out;
    out <== 1 + in - 2*in;
}
template NAND() {
    signal input a;
    signal input b;
    signal output out;
    out <== 1 - a*b;
}
template NOR() {
    signal input a;
    signal input b;
    signal output out;
    out <== a*b + 1 - a - b;
}
template MultiAND(n) {
    signal input in[n];
    signal output out;
    component and1;
    component and2;
    component ands[2];
    if (n==1) {
        out <== in[0];
    } else if (n==2) {
        and1 = AND();
        and1.a <== in[0];
        and1.b <== in[1];
        out <== and1.out;
    } else {
        and2 = AND();
        var n1 = n\2;
        var n2 = n-n\2;
        ands[0] = MultiAND(n1);
        ands[1] = MultiAND(n2);
        var i;
        for (i=0; i<n1; i++) ands[0].in[i] <== in[i];
        for (i=0; i<n2; i++) ands[1].in[i] <== in[n1+i];
        and2.a <== ands[0].out;
        and2.b <== ands[1].out;
        out <== and2.out;
    }
}
)CIRCOM";
  if (base == "binsum.circom" || base == "aliascheck.circom" || base == "compconstant.circom" || base == "sign.circom" ||
      base == "mux1.circom")
    return "\n";
  return nullptr;
}

// Load `tname(args...)` from `path`; include directories separated by ':'.
// chain_limit: -1 = both recurrences from scan tables where possible (ZKWG_NET_CHAIN overrides), 0 = every gate in the list (what
// the load-time self-check of zkwg_net_host.h compares the tables with), 1 = the forward recurrence only
static inline bool load(const std::string& path, const std::string& include_dirs, const std::string& tname,
                        const std::vector<i64>& args, Net& net, std::string& err, int chain_limit = -1) {
  try {
    Elab E;
    E.chain_limit = chain_limit;
    std::stringstream ss(include_dirs);
    std::string d;
    while (std::getline(ss, d, ':')) if (!d.empty()) E.include_dirs.push_back(d);
    E.build(path, tname, args, net);
    return true;
  } catch (Error& e) {
    err = e.msg;
    return false;
  } catch (std::exception& e) {
    err = e.what();
    return false;
  }
}

}  // namespace zkc
