// lua_subset.cpp -- see lua_subset.h.
#include "lua_subset.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace tdl {

// ------------------------------------------------------------------------------------------------
// lexer
// ------------------------------------------------------------------------------------------------
enum Tok { T_EOF, T_NAME, T_INT, T_FLT, T_STR, T_SYM, T_KW };
struct Token {
    Tok t = T_EOF;
    std::string s;
    long long i = 0;
    double d = 0;
    int line = 1;
};

static const char* kKeywords[] = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "goto",
                                  "if", "in", "local", "nil", "not", "or", "repeat", "return", "then", "true",
                                  "until", "while"};

static bool is_kw(const std::string& s) {
    for (auto k : kKeywords)
        if (s == k) return true;
    return false;
}

static std::vector<Token> lex(const std::string& src) {
    std::vector<Token> out;
    size_t p = 0;
    int line = 1;
    auto err = [&](const std::string& m) { throw LuaError{"line " + std::to_string(line) + ": " + m}; };
    auto long_bracket = [&](size_t at, size_t* body, size_t* end) -> bool {   // [[ ... ]] / [=[ ... ]=]
        if (src[at] != '[') return false;
        size_t q = at + 1, eq = 0;
        while (q < src.size() && src[q] == '=') { ++q; ++eq; }
        if (q >= src.size() || src[q] != '[') return false;
        std::string close = "]" + std::string(eq, '=') + "]";
        size_t e = src.find(close, q + 1);
        if (e == std::string::npos) err("unfinished long bracket");
        *body = q + 1;
        *end = e + close.size();
        return true;
    };
    while (p < src.size()) {
        char c = src[p];
        if (c == '\n') { ++line; ++p; continue; }
        if (c == ' ' || c == '\t' || c == '\r') { ++p; continue; }
        if (c == '-' && p + 1 < src.size() && src[p + 1] == '-') {
            size_t body, end;
            if (p + 2 < src.size() && long_bracket(p + 2, &body, &end)) {
                for (size_t q = p; q < end; ++q) if (src[q] == '\n') ++line;
                p = end;
            } else {
                while (p < src.size() && src[p] != '\n') ++p;
            }
            continue;
        }
        Token t;
        t.line = line;
        if (isalpha((unsigned char)c) || c == '_') {
            size_t q = p;
            while (q < src.size() && (isalnum((unsigned char)src[q]) || src[q] == '_')) ++q;
            t.s = src.substr(p, q - p);
            t.t = is_kw(t.s) ? T_KW : T_NAME;
            p = q;
        } else if (isdigit((unsigned char)c) || (c == '.' && p + 1 < src.size() && isdigit((unsigned char)src[p + 1]))) {
            size_t q = p;
            bool is_float = false;
            if (c == '0' && q + 1 < src.size() && (src[q + 1] == 'x' || src[q + 1] == 'X')) {
                q += 2;
                while (q < src.size() && (isxdigit((unsigned char)src[q]) || src[q] == '.' || src[q] == 'p' || src[q] == 'P' ||
                                          ((src[q] == '+' || src[q] == '-') && (src[q - 1] == 'p' || src[q - 1] == 'P')))) {
                    if (src[q] == '.' || src[q] == 'p' || src[q] == 'P') is_float = true;
                    ++q;
                }
            } else {
                while (q < src.size() && (isdigit((unsigned char)src[q]) || src[q] == '.' || src[q] == 'e' || src[q] == 'E' ||
                                          ((src[q] == '+' || src[q] == '-') && (src[q - 1] == 'e' || src[q - 1] == 'E')))) {
                    if (src[q] == '.' || src[q] == 'e' || src[q] == 'E') is_float = true;
                    ++q;
                }
            }
            std::string num = src.substr(p, q - p);
            if (is_float) {
                t.t = T_FLT;
                t.d = strtod(num.c_str(), nullptr);
            } else {
                t.t = T_INT;
                t.i = (long long)strtoull(num.c_str(), nullptr, 0);
            }
            p = q;
        } else if (c == '"' || c == '\'') {
            size_t q = p + 1;
            std::string s;
            while (q < src.size() && src[q] != c) {
                if (src[q] == '\n') err("unfinished string");
                if (src[q] == '\\' && q + 1 < src.size()) {
                    char e = src[q + 1];
                    q += 2;
                    switch (e) {
                        case 'n': s += '\n'; break;
                        case 't': s += '\t'; break;
                        case 'r': s += '\r'; break;
                        case '0': s += '\0'; break;
                        case '\\': s += '\\'; break;
                        case '"': s += '"'; break;
                        case '\'': s += '\''; break;
                        case '\n': s += '\n'; ++line; break;
                        default: err(std::string("unsupported escape \\") + e);
                    }
                } else {
                    s += src[q++];
                }
            }
            if (q >= src.size()) err("unfinished string");
            t.t = T_STR;
            t.s = s;
            p = q + 1;
        } else if (c == '[') {
            size_t body, end;
            if (long_bracket(p, &body, &end)) {
                size_t b = body;
                if (b < src.size() && src[b] == '\n') ++b;
                size_t close_len = end - src.rfind(']', end - 2);   // length of the closing bracket
                t.t = T_STR;
                t.s = src.substr(b, end - close_len - b);
                for (size_t q = p; q < end; ++q) if (src[q] == '\n') ++line;
                p = end;
            } else {
                t.t = T_SYM;
                t.s = "[";
                ++p;
            }
        } else {
            static const char* syms[] = {"...", "..", "==", "~=", "<=", ">=", "//", "::", "<<", ">>", "+", "-", "*", "/", "%",
                                         "^", "#", "&", "~", "|", "<", ">", "=", "(", ")", "{", "}", "]", ";", ":", ",", "."};
            bool found = false;
            for (auto sy : syms) {
                size_t n = strlen(sy);
                if (src.compare(p, n, sy) == 0) {
                    t.t = T_SYM;
                    t.s = sy;
                    p += n;
                    found = true;
                    break;
                }
            }
            if (!found) err(std::string("unexpected character '") + c + "'");
        }
        out.push_back(t);
    }
    Token e;
    e.line = line;
    out.push_back(e);
    return out;
}

// ------------------------------------------------------------------------------------------------
// AST
// ------------------------------------------------------------------------------------------------
struct Expr;
using ExprP = std::shared_ptr<Expr>;
struct Expr {
    enum K { CONST, NAME, INDEX, CALL, BIN, UN, TABLE } k = CONST;
    int line = 0;
    Value v;                       // CONST
    std::string name;              // NAME; BIN/UN operator
    ExprP a, b;                    // INDEX (a[b]); BIN; UN (a); CALL callee (a)
    std::vector<ExprP> args;       // CALL args; TABLE positional
    std::vector<std::pair<ExprP, ExprP>> fields;   // TABLE keyed
};
struct Stmt;
using StmtP = std::shared_ptr<Stmt>;
using Block = std::vector<StmtP>;
struct Stmt {
    enum K { ASSIGN, LOCAL, CALL, FORNUM, FORIN, WHILE, IF, DO, BREAK } k = CALL;
    int line = 0;
    std::vector<ExprP> lhs, rhs;                 // ASSIGN / LOCAL (lhs names as NAME exprs)
    ExprP e;                                     // CALL expr; WHILE cond; FORIN iterable
    std::string var, var2;                       // FORNUM / FORIN
    ExprP from, to, step;                        // FORNUM
    std::vector<std::pair<ExprP, Block>> arms;   // IF arms (cond == nullptr -> else)
    Block body;
};

struct Parser {
    std::vector<Token> toks;
    size_t p = 0;
    const Token& cur() const { return toks[p]; }
    [[noreturn]] void err(const std::string& m) const { throw LuaError{"line " + std::to_string(cur().line) + ": " + m}; }
    bool is_sym(const char* s) const { return cur().t == T_SYM && cur().s == s; }
    bool is_kw(const char* s) const { return cur().t == T_KW && cur().s == s; }
    bool accept_sym(const char* s) { if (is_sym(s)) { ++p; return true; } return false; }
    bool accept_kw(const char* s) { if (is_kw(s)) { ++p; return true; } return false; }
    void expect_sym(const char* s) { if (!accept_sym(s)) err(std::string("'") + s + "' expected near '" + cur().s + "'"); }
    void expect_kw(const char* s) { if (!accept_kw(s)) err(std::string("'") + s + "' expected near '" + cur().s + "'"); }
    std::string expect_name() {
        if (cur().t != T_NAME) err("name expected near '" + cur().s + "'");
        return toks[p++].s;
    }

    ExprP mk(Expr::K k) { auto e = std::make_shared<Expr>(); e->k = k; e->line = cur().line; return e; }

    ExprP primary() {
        ExprP e;
        if (cur().t == T_NAME) {
            e = mk(Expr::NAME);
            e->name = toks[p++].s;
        } else if (accept_sym("(")) {
            e = expr(0);
            expect_sym(")");
        } else {
            err("unexpected symbol near '" + cur().s + "'");
        }
        for (;;) {
            if (accept_sym(".")) {
                auto i = mk(Expr::INDEX);
                i->a = e;
                i->b = mk(Expr::CONST);
                i->b->v = Value::string(expect_name());
                e = i;
            } else if (accept_sym("[")) {
                auto i = mk(Expr::INDEX);
                i->a = e;
                i->b = expr(0);
                expect_sym("]");
                e = i;
            } else if (is_sym("(")) {
                ++p;
                auto c = mk(Expr::CALL);
                c->a = e;
                if (!is_sym(")")) {
                    do c->args.push_back(expr(0)); while (accept_sym(","));
                }
                expect_sym(")");
                e = c;
            } else if (cur().t == T_STR) {   // f"str"
                auto c = mk(Expr::CALL);
                c->a = e;
                auto s = mk(Expr::CONST);
                s->v = Value::string(toks[p++].s);
                c->args.push_back(s);
                e = c;
            } else if (is_sym("{")) {   // f{...}
                auto c = mk(Expr::CALL);
                c->a = e;
                c->args.push_back(table());
                e = c;
            } else if (is_sym(":")) {
                err("method calls are not supported");
            } else {
                return e;
            }
        }
    }
    ExprP table() {
        expect_sym("{");
        auto t = mk(Expr::TABLE);
        while (!is_sym("}")) {
            if (cur().t == T_NAME && toks[p + 1].t == T_SYM && toks[p + 1].s == "=") {
                auto k = mk(Expr::CONST);
                k->v = Value::string(toks[p].s);
                p += 2;
                t->fields.push_back({k, expr(0)});
            } else if (accept_sym("[")) {
                auto k = expr(0);
                expect_sym("]");
                expect_sym("=");
                t->fields.push_back({k, expr(0)});
            } else {
                t->args.push_back(expr(0));
            }
            if (!accept_sym(",") && !accept_sym(";")) break;
        }
        expect_sym("}");
        return t;
    }
    ExprP simple() {
        const Token& t = cur();
        if (t.t == T_INT) { auto e = mk(Expr::CONST); e->v = Value::integer(t.i); ++p; return e; }
        if (t.t == T_FLT) { auto e = mk(Expr::CONST); e->v = Value::number(t.d); ++p; return e; }
        if (t.t == T_STR) { auto e = mk(Expr::CONST); e->v = Value::string(t.s); ++p; return e; }
        if (is_kw("nil")) { ++p; return mk(Expr::CONST); }
        if (is_kw("true")) { auto e = mk(Expr::CONST); e->v = Value::boolean(true); ++p; return e; }
        if (is_kw("false")) { auto e = mk(Expr::CONST); e->v = Value::boolean(false); ++p; return e; }
        if (is_kw("function")) err("function definitions are not supported by this front-end");
        if (is_sym("{")) return table();
        return primary();
    }
    // Lua operator precedences (left, right)
    static bool binprec(const Token& t, int* l, int* r) {
        static const struct { const char* op; int l, r; } tab[] = {
            {"or", 1, 1}, {"and", 2, 2}, {"<", 3, 3}, {">", 3, 3}, {"<=", 3, 3}, {">=", 3, 3}, {"~=", 3, 3}, {"==", 3, 3},
            {"..", 9, 8}, {"+", 10, 10}, {"-", 10, 10}, {"*", 11, 11}, {"/", 11, 11}, {"//", 11, 11}, {"%", 11, 11},
            {"^", 14, 13}};
        if (t.t != T_SYM && t.t != T_KW) return false;
        for (auto& e : tab)
            if (t.s == e.op) { *l = e.l; *r = e.r; return true; }
        return false;
    }
    ExprP expr(int limit) {
        ExprP e;
        if (is_kw("not") || is_sym("-") || is_sym("#")) {
            auto u = mk(Expr::UN);
            u->name = toks[p++].s;
            u->a = expr(12);
            e = u;
        } else {
            e = simple();
        }
        int l, r;
        while (binprec(cur(), &l, &r) && l > limit) {
            auto b = mk(Expr::BIN);
            b->name = toks[p++].s;
            b->a = e;
            b->b = expr(r);
            e = b;
        }
        return e;
    }

    Block block() {
        Block b;
        for (;;) {
            while (accept_sym(";")) {}
            if (cur().t == T_EOF || is_kw("end") || is_kw("else") || is_kw("elseif") || is_kw("until")) return b;
            b.push_back(statement());
        }
    }
    StmtP statement() {
        auto s = std::make_shared<Stmt>();
        s->line = cur().line;
        if (accept_kw("local")) {
            if (is_kw("function")) err("function definitions are not supported by this front-end");
            s->k = Stmt::LOCAL;
            do { auto n = mk(Expr::NAME); n->name = expect_name(); s->lhs.push_back(n); } while (accept_sym(","));
            if (accept_sym("=")) do s->rhs.push_back(expr(0)); while (accept_sym(","));
            return s;
        }
        if (accept_kw("for")) {
            s->var = expect_name();
            if (accept_sym("=")) {
                s->k = Stmt::FORNUM;
                s->from = expr(0);
                expect_sym(",");
                s->to = expr(0);
                if (accept_sym(",")) s->step = expr(0);
            } else {
                s->k = Stmt::FORIN;
                if (accept_sym(",")) s->var2 = expect_name();
                expect_kw("in");
                s->e = expr(0);
            }
            expect_kw("do");
            s->body = block();
            expect_kw("end");
            return s;
        }
        if (accept_kw("while")) {
            s->k = Stmt::WHILE;
            s->e = expr(0);
            expect_kw("do");
            s->body = block();
            expect_kw("end");
            return s;
        }
        if (accept_kw("if")) {
            s->k = Stmt::IF;
            ExprP c = expr(0);
            expect_kw("then");
            s->arms.push_back({c, block()});
            for (;;) {
                if (accept_kw("elseif")) {
                    ExprP c2 = expr(0);
                    expect_kw("then");
                    s->arms.push_back({c2, block()});
                } else if (accept_kw("else")) {
                    s->arms.push_back({nullptr, block()});
                    expect_kw("end");
                    break;
                } else {
                    expect_kw("end");
                    break;
                }
            }
            return s;
        }
        if (accept_kw("do")) {
            s->k = Stmt::DO;
            s->body = block();
            expect_kw("end");
            return s;
        }
        if (accept_kw("break")) { s->k = Stmt::BREAK; return s; }
        if (is_kw("function") || is_kw("return") || is_kw("repeat") || is_kw("goto"))
            err("'" + cur().s + "' is not supported by this front-end");
        ExprP e = primary();
        if (is_sym("=") || is_sym(",")) {
            s->k = Stmt::ASSIGN;
            s->lhs.push_back(e);
            while (accept_sym(",")) s->lhs.push_back(primary());
            expect_sym("=");
            do s->rhs.push_back(expr(0)); while (accept_sym(","));
            for (auto& l : s->lhs)
                if (l->k != Expr::NAME && l->k != Expr::INDEX) err("cannot assign to this expression");
            return s;
        }
        if (e->k != Expr::CALL) err("syntax error near '" + cur().s + "'");
        s->k = Stmt::CALL;
        s->e = e;
        return s;
    }
};

// ------------------------------------------------------------------------------------------------
// evaluation
// ------------------------------------------------------------------------------------------------
static std::string fmt_float(double d) {   // lua_Number2str "%.14g" + ".0" for integral floats
    if (d != d) return d < 0 ? "-nan" : "nan";
    if (isinf(d)) return d < 0 ? "-inf" : "inf";
    char buf[64];
    snprintf(buf, sizeof buf, "%.14g", d);
    if (!strpbrk(buf, ".eEn")) strcat(buf, ".0");
    return buf;
}
std::string tostring(const Value& v) {
    switch (v.type) {
        case Value::NIL: return "nil";
        case Value::BOOL: return v.b ? "true" : "false";
        case Value::INT: return std::to_string(v.i);
        case Value::FLT: return fmt_float(v.d);
        case Value::STR: return v.s;
        case Value::TAB: return "table";
        default: return "function";
    }
}

static bool tonumber(const Value& v, Value* out) {
    if (v.is_number()) { *out = v; return true; }
    if (v.type == Value::STR) {
        const char* s = v.s.c_str();
        char* end = nullptr;
        while (isspace((unsigned char)*s)) ++s;
        if (!*s) return false;
        long long i = strtoll(s, &end, 0);
        while (end && isspace((unsigned char)*end)) ++end;
        if (end && !*end && !strpbrk(s, ".eEpPnN")) { *out = Value::integer(i); return true; }
        double d = strtod(s, &end);
        while (end && isspace((unsigned char)*end)) ++end;
        if (end && !*end) { *out = Value::number(d); return true; }
    }
    return false;
}

struct Exec {
    Interp& in;
    std::vector<std::map<std::string, Value>> scopes;
    struct BreakSignal {};
    long budget = 50000000;   // statement budget: a runaway script fails instead of hanging the loader

    explicit Exec(Interp& i) : in(i) {}
    [[noreturn]] void err(int line, const std::string& m) { throw LuaError{"line " + std::to_string(line) + ": " + m}; }

    Value* find(const std::string& n) {
        for (size_t k = scopes.size(); k-- > 0;) {
            auto it = scopes[k].find(n);
            if (it != scopes[k].end()) return &it->second;
        }
        auto it = in.globals.find(n);
        return it == in.globals.end() ? nullptr : &it->second;
    }

    Value arith(int line, const std::string& op, const Value& a0, const Value& b0) {
        Value a, b;
        if (!tonumber(a0, &a) || !tonumber(b0, &b))
            err(line, "attempt to perform arithmetic on a " + std::string(a0.is_number() || a0.type == Value::STR ? "" : "non-number ") + "value");
        const bool ints = a.type == Value::INT && b.type == Value::INT;
        if (op == "+") return ints ? Value::integer((long long)((unsigned long long)a.i + (unsigned long long)b.i)) : Value::number(a.as_double() + b.as_double());
        if (op == "-") return ints ? Value::integer((long long)((unsigned long long)a.i - (unsigned long long)b.i)) : Value::number(a.as_double() - b.as_double());
        if (op == "*") return ints ? Value::integer((long long)((unsigned long long)a.i * (unsigned long long)b.i)) : Value::number(a.as_double() * b.as_double());
        if (op == "/") return Value::number(a.as_double() / b.as_double());
        if (op == "^") return Value::number(pow(a.as_double(), b.as_double()));
        if (op == "//") {
            if (ints) {
                if (b.i == 0) err(line, "attempt to perform 'n//0'");
                long long q = a.i / b.i;
                if ((a.i % b.i != 0) && ((a.i < 0) != (b.i < 0))) --q;
                return Value::integer(q);
            }
            return Value::number(floor(a.as_double() / b.as_double()));
        }
        if (op == "%") {
            if (ints) {
                if (b.i == 0) err(line, "attempt to perform 'n%%0'");
                long long m = a.i % b.i;
                if (m != 0 && ((m < 0) != (b.i < 0))) m += b.i;
                return Value::integer(m);
            }
            double x = a.as_double(), y = b.as_double();
            double m = fmod(x, y);
            if (m != 0 && ((m < 0) != (y < 0))) m += y;
            return Value::number(m);
        }
        err(line, "unknown operator " + op);
    }
    static bool raw_equal(const Value& a, const Value& b) {
        if (a.is_number() && b.is_number()) {
            if (a.type == Value::INT && b.type == Value::INT) return a.i == b.i;
            return a.as_double() == b.as_double();
        }
        if (a.type != b.type) return false;
        switch (a.type) {
            case Value::NIL: return true;
            case Value::BOOL: return a.b == b.b;
            case Value::STR: return a.s == b.s;
            case Value::TAB: return a.tab == b.tab;
            default: return a.s == b.s;
        }
    }
    bool less(int line, const Value& a, const Value& b, bool or_equal) {
        if (a.is_number() && b.is_number()) {
            if (a.type == Value::INT && b.type == Value::INT) return or_equal ? a.i <= b.i : a.i < b.i;
            return or_equal ? a.as_double() <= b.as_double() : a.as_double() < b.as_double();
        }
        if (a.type == Value::STR && b.type == Value::STR) return or_equal ? a.s <= b.s : a.s < b.s;
        err(line, "attempt to compare incompatible values");
    }

    Value index(int line, const Value& t, const Value& k) {
        if (t.type != Value::TAB) err(line, "attempt to index a " + std::string(t.type == Value::NIL ? "nil" : "non-table") + " value");
        Value kn;
        if (k.type == Value::FLT && k.d == floor(k.d)) kn = Value::integer((long long)k.d); else kn = k;
        if (kn.type == Value::INT) {
            if (kn.i >= 1 && (size_t)kn.i <= t.tab->arr.size()) return t.tab->arr[(size_t)kn.i - 1];
            auto it = t.tab->hash.find("#" + std::to_string(kn.i));
            return it == t.tab->hash.end() ? Value::nil() : it->second;
        }
        if (kn.type == Value::STR) {
            auto it = t.tab->hash.find(kn.s);
            return it == t.tab->hash.end() ? Value::nil() : it->second;
        }
        return Value::nil();
    }
    void store(int line, Value& t, const Value& k, const Value& v) {
        if (t.type != Value::TAB) err(line, "attempt to index a non-table value");
        Value kn;
        if (k.type == Value::FLT && k.d == floor(k.d)) kn = Value::integer((long long)k.d); else kn = k;
        if (kn.type == Value::INT) {
            auto& arr = t.tab->arr;
            if (kn.i >= 1 && (size_t)kn.i <= arr.size()) { arr[(size_t)kn.i - 1] = v; return; }
            if (kn.i >= 1 && (size_t)kn.i == arr.size() + 1) { arr.push_back(v); return; }
            t.tab->hash["#" + std::to_string(kn.i)] = v;
            return;
        }
        if (kn.type == Value::STR) { t.tab->hash[kn.s] = v; return; }
        err(line, "unsupported table key");
    }

    Value call(int line, const Expr& c) {
        // callee: plain global name or lib.field
        std::string fname;
        if (c.a->k == Expr::NAME) fname = c.a->name;
        else if (c.a->k == Expr::INDEX && c.a->a->k == Expr::NAME && c.a->b->k == Expr::CONST && c.a->b->v.type == Value::STR)
            fname = c.a->a->name + "." + c.a->b->v.s;
        else err(line, "unsupported call target");
        std::vector<Value> args;
        for (auto& a : c.args) args.push_back(eval(*a));
        auto it = in.fns_.find(fname);
        if (it == in.fns_.end()) err(line, "attempt to call a nil value (global '" + fname + "')");
        try {
            return it->second(args);
        } catch (LuaError& e) {
            err(line, e.msg);
        }
    }

    Value eval(const Expr& e) {
        switch (e.k) {
            case Expr::CONST: return e.v;
            case Expr::NAME: {
                Value* v = find(e.name);
                return v ? *v : Value::nil();
            }
            case Expr::INDEX: {
                if (e.a->k == Expr::NAME && !find(e.a->name) && e.b->k == Expr::CONST && e.b->v.type == Value::STR) {
                    // library constants such as math.pi / math.huge
                    const std::string q = e.a->name + "." + e.b->v.s;
                    if (q == "math.pi") return Value::number(3.14159265358979323846);
                    if (q == "math.huge") return Value::number(HUGE_VAL);
                    if (q == "math.maxinteger") return Value::integer(9223372036854775807LL);
                    if (q == "math.mininteger") return Value::integer((long long)(-9223372036854775807LL - 1));
                }
                return index(e.line, eval(*e.a), eval(*e.b));
            }
            case Expr::CALL: return call(e.line, e);
            case Expr::TABLE: {
                Value t;
                t.type = Value::TAB;
                t.tab = std::make_shared<Table>();
                for (auto& a : e.args) t.tab->arr.push_back(eval(*a));
                for (auto& f : e.fields) store(e.line, t, eval(*f.first), eval(*f.second));
                return t;
            }
            case Expr::UN: {
                Value a = eval(*e.a);
                if (e.name == "not") return Value::boolean(!a.truthy());
                if (e.name == "#") {
                    if (a.type == Value::STR) return Value::integer((long long)a.s.size());
                    if (a.type == Value::TAB) return Value::integer((long long)a.tab->arr.size());
                    err(e.line, "attempt to get length of a non-table value");
                }
                Value n;
                if (!tonumber(a, &n)) err(e.line, "attempt to perform arithmetic on a non-number value");
                return n.type == Value::INT ? Value::integer((long long)(0ull - (unsigned long long)n.i)) : Value::number(-n.d);
            }
            case Expr::BIN: {
                const std::string& op = e.name;
                if (op == "and") { Value a = eval(*e.a); return a.truthy() ? eval(*e.b) : a; }
                if (op == "or") { Value a = eval(*e.a); return a.truthy() ? a : eval(*e.b); }
                Value a = eval(*e.a), b = eval(*e.b);
                if (op == "..") {
                    if ((a.type != Value::STR && !a.is_number()) || (b.type != Value::STR && !b.is_number()))
                        err(e.line, "attempt to concatenate a non-string value");
                    return Value::string(tostring(a) + tostring(b));
                }
                if (op == "==") return Value::boolean(raw_equal(a, b));
                if (op == "~=") return Value::boolean(!raw_equal(a, b));
                if (op == "<") return Value::boolean(less(e.line, a, b, false));
                if (op == "<=") return Value::boolean(less(e.line, a, b, true));
                if (op == ">") return Value::boolean(less(e.line, b, a, false));
                if (op == ">=") return Value::boolean(less(e.line, b, a, true));
                return arith(e.line, op, a, b);
            }
        }
        return Value::nil();
    }

    void assign(int line, const Expr& target, const Value& v, bool local) {
        if (target.k == Expr::NAME) {
            if (local) { scopes.back()[target.name] = v; return; }
            Value* slot = find(target.name);
            if (slot) *slot = v; else in.globals[target.name] = v;
            return;
        }
        Value t = eval(*target.a);   // tables are shared_ptr: storing through the copy mutates the table
        store(line, t, eval(*target.b), v);
    }

    void run_block(const Block& b) {
        scopes.emplace_back();
        struct Pop { std::vector<std::map<std::string, Value>>& s; ~Pop() { s.pop_back(); } } pop{scopes};
        for (auto& s : b) run_stmt(*s);
    }
    void run_stmt(const Stmt& s) {
        if (--budget < 0) err(s.line, "script exceeded the statement budget");
        switch (s.k) {
            case Stmt::CALL: eval(*s.e); break;
            case Stmt::ASSIGN:
            case Stmt::LOCAL: {
                std::vector<Value> vals;
                for (auto& r : s.rhs) vals.push_back(eval(*r));
                for (size_t i = 0; i < s.lhs.size(); ++i)
                    assign(s.line, *s.lhs[i], i < vals.size() ? vals[i] : Value::nil(), s.k == Stmt::LOCAL);
            } break;
            case Stmt::DO: run_block(s.body); break;
            case Stmt::BREAK: throw BreakSignal{};
            case Stmt::IF:
                for (auto& arm : s.arms)
                    if (!arm.first || eval(*arm.first).truthy()) { run_block(arm.second); break; }
                break;
            case Stmt::WHILE:
                try {
                    while (eval(*s.e).truthy()) {
                        if (--budget < 0) err(s.line, "script exceeded the statement budget");
                        run_block(s.body);
                    }
                } catch (BreakSignal&) {}
                break;
            case Stmt::FORNUM: {
                Value a, b, c = Value::integer(1);
                if (!tonumber(eval(*s.from), &a)) err(s.line, "'for' initial value must be a number");
                if (!tonumber(eval(*s.to), &b)) err(s.line, "'for' limit must be a number");
                if (s.step && !tonumber(eval(*s.step), &c)) err(s.line, "'for' step must be a number");
                try {
                    if (a.type == Value::INT && c.type == Value::INT) {
                        if (c.i == 0) err(s.line, "'for' step is zero");
                        long long lim = b.type == Value::INT ? b.i : (long long)(c.i > 0 ? floor(b.d) : ceil(b.d));
                        for (long long i = a.i; c.i > 0 ? i <= lim : i >= lim; i += c.i) {
                            scopes.emplace_back();
                            scopes.back()[s.var] = Value::integer(i);
                            struct Pop { std::vector<std::map<std::string, Value>>& s; ~Pop() { s.pop_back(); } } pop{scopes};
                            run_block(s.body);
                            if (--budget < 0) err(s.line, "script exceeded the statement budget");
                        }
                    } else {
                        const double st = c.as_double(), lim = b.as_double();
                        if (st == 0) err(s.line, "'for' step is zero");
                        for (double i = a.as_double(); st > 0 ? i <= lim : i >= lim; i += st) {
                            scopes.emplace_back();
                            scopes.back()[s.var] = Value::number(i);
                            struct Pop { std::vector<std::map<std::string, Value>>& s; ~Pop() { s.pop_back(); } } pop{scopes};
                            run_block(s.body);
                            if (--budget < 0) err(s.line, "script exceeded the statement budget");
                        }
                    }
                } catch (BreakSignal&) {}
            } break;
            case Stmt::FORIN: {
                // only `for i, v in ipairs(t)` / `for k, v in pairs(t)` (array part, then named fields)
                if (s.e->k != Expr::CALL || s.e->a->k != Expr::NAME || (s.e->a->name != "ipairs" && s.e->a->name != "pairs") ||
                    s.e->args.size() != 1)
                    err(s.line, "only 'for .. in ipairs(t)' / 'pairs(t)' loops are supported");
                Value t = eval(*s.e->args[0]);
                if (t.type != Value::TAB) err(s.line, "bad argument #1 to 'ipairs' (table expected)");
                try {
                    for (size_t i = 0; i < t.tab->arr.size(); ++i) {
                        if (s.e->a->name == "ipairs" && t.tab->arr[i].type == Value::NIL) break;
                        scopes.emplace_back();
                        scopes.back()[s.var] = Value::integer((long long)i + 1);
                        if (!s.var2.empty()) scopes.back()[s.var2] = t.tab->arr[i];
                        struct Pop { std::vector<std::map<std::string, Value>>& s; ~Pop() { s.pop_back(); } } pop{scopes};
                        run_block(s.body);
                    }
                    if (s.e->a->name == "pairs")
                        for (auto& kv : t.tab->hash) {
                            scopes.emplace_back();
                            scopes.back()[s.var] = Value::string(kv.first);
                            if (!s.var2.empty()) scopes.back()[s.var2] = kv.second;
                            struct Pop { std::vector<std::map<std::string, Value>>& s; ~Pop() { s.pop_back(); } } pop{scopes};
                            run_block(s.body);
                        }
                } catch (BreakSignal&) {}
            } break;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------------
static double num_arg(const std::vector<Value>& a, size_t i, const char* fn) {
    Value n;
    if (i >= a.size() || !tonumber(a[i], &n)) throw LuaError{std::string("bad argument #") + std::to_string(i + 1) + " to '" + fn + "' (number expected)"};
    return n.as_double();
}

static Value lib_format(const std::vector<Value>& a) {
    if (a.empty() || a[0].type != Value::STR) throw LuaError{"bad argument #1 to 'format' (string expected)"};
    const std::string& f = a[0].s;
    std::string out;
    size_t arg = 1;
    for (size_t p = 0; p < f.size(); ++p) {
        if (f[p] != '%') { out += f[p]; continue; }
        if (p + 1 < f.size() && f[p + 1] == '%') { out += '%'; ++p; continue; }
        size_t q = p + 1;
        while (q < f.size() && strchr("-+ #0123456789.", f[q])) ++q;
        if (q >= f.size()) throw LuaError{"invalid format string to 'format'"};
        std::string spec = f.substr(p, q - p);
        const char conv = f[q];
        char buf[512];
        if (arg >= a.size()) throw LuaError{"bad argument #" + std::to_string(arg + 1) + " to 'format' (no value)"};
        if (conv == 'd' || conv == 'i') {
            Value n;
            if (!tonumber(a[arg], &n)) throw LuaError{"bad argument to 'format' (number expected)"};
            if (n.type == Value::FLT && n.d != floor(n.d)) throw LuaError{"bad argument to 'format' (number has no integer representation)"};
            snprintf(buf, sizeof buf, (spec + "lld").c_str(), n.type == Value::INT ? n.i : (long long)n.d);
        } else if (strchr("fFgGeE", conv)) {
            snprintf(buf, sizeof buf, (spec + conv).c_str(), num_arg(a, arg, "format"));
        } else if (conv == 'x' || conv == 'X') {
            snprintf(buf, sizeof buf, (spec + "ll" + conv).c_str(), (long long)num_arg(a, arg, "format"));
        } else if (conv == 's') {
            snprintf(buf, sizeof buf, (spec + "s").c_str(), tostring(a[arg]).c_str());
        } else {
            throw LuaError{std::string("invalid conversion '%") + conv + "' to 'format'"};
        }
        out += buf;
        ++arg;
        p = q;
    }
    return Value::string(out);
}

Interp::Interp() {
    fns_["string.format"] = lib_format;
    fns_["tostring"] = [](const std::vector<Value>& a) { return Value::string(a.empty() ? "nil" : tostring(a[0])); };
    fns_["tonumber"] = [](const std::vector<Value>& a) {
        Value n;
        return (!a.empty() && tonumber(a[0], &n)) ? n : Value::nil();
    };
    fns_["print"] = [](const std::vector<Value>&) { return Value::nil(); };
    fns_["math.floor"] = [](const std::vector<Value>& a) {
        if (!a.empty() && a[0].type == Value::INT) return a[0];
        return Value::integer((long long)floor(num_arg(a, 0, "floor")));
    };
    fns_["math.ceil"] = [](const std::vector<Value>& a) {
        if (!a.empty() && a[0].type == Value::INT) return a[0];
        return Value::integer((long long)ceil(num_arg(a, 0, "ceil")));
    };
    fns_["math.abs"] = [](const std::vector<Value>& a) {
        if (!a.empty() && a[0].type == Value::INT) return Value::integer(a[0].i < 0 ? -a[0].i : a[0].i);
        return Value::number(fabs(num_arg(a, 0, "abs")));
    };
    fns_["math.sqrt"] = [](const std::vector<Value>& a) { return Value::number(sqrt(num_arg(a, 0, "sqrt"))); };
    fns_["math.sin"] = [](const std::vector<Value>& a) { return Value::number(sin(num_arg(a, 0, "sin"))); };
    fns_["math.cos"] = [](const std::vector<Value>& a) { return Value::number(cos(num_arg(a, 0, "cos"))); };
    fns_["math.exp"] = [](const std::vector<Value>& a) { return Value::number(exp(num_arg(a, 0, "exp"))); };
    fns_["math.log"] = [](const std::vector<Value>& a) { return Value::number(log(num_arg(a, 0, "log"))); };
    fns_["math.fmod"] = [](const std::vector<Value>& a) { return Value::number(fmod(num_arg(a, 0, "fmod"), num_arg(a, 1, "fmod"))); };
    fns_["math.max"] = [](const std::vector<Value>& a) {
        if (a.empty()) throw LuaError{"bad argument #1 to 'max' (number expected)"};
        size_t best = 0;
        for (size_t i = 1; i < a.size(); ++i)
            if (num_arg(a, i, "max") > num_arg(a, best, "max")) best = i;
        return a[best];
    };
    fns_["math.min"] = [](const std::vector<Value>& a) {
        if (a.empty()) throw LuaError{"bad argument #1 to 'min' (number expected)"};
        size_t best = 0;
        for (size_t i = 1; i < a.size(); ++i)
            if (num_arg(a, i, "min") < num_arg(a, best, "min")) best = i;
        return a[best];
    };
}

void Interp::set_function(const std::string& name, HostFn fn) { fns_[name] = std::move(fn); }

bool Interp::run(const std::string& source, std::string* err) {
    try {
        Parser ps;
        ps.toks = lex(source);
        Block b = ps.block();
        if (ps.cur().t != T_EOF) ps.err("'<eof>' expected near '" + ps.cur().s + "'");
        Exec ex(*this);
        try {
            ex.run_block(b);
        } catch (Exec::BreakSignal&) {
            throw LuaError{"break outside a loop"};
        }
        return true;
    } catch (LuaError& e) {
        *err = e.msg;
        return false;
    }
}

}  // namespace tdl
