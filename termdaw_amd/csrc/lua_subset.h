// lua_subset.h -- a small interpreter for the subset of Lua 5.3 that termdaw project scripts use.
//
// The reference embeds a full Lua 5.3 through the mlua crate (state.rs:17,83-159); its scripts only
// *record* calls of 23 registered globals (state.rs:103-157).  No Lua exists in this image, so the
// project front-end carries its own interpreter for: comments, global/local assignment, function-call
// statements, numeric `for`, `for .. in ipairs()`, `while`, `if/elseif/else`, numbers (integer and
// float subtypes), strings, booleans, nil, table constructors with positional and named fields,
// indexing, the arithmetic / comparison / logical / concatenation / length operators with Lua's
// precedences, and a few library functions (string.format, tostring, tonumber, math.*).
// `function` definitions, metatables, coroutines and goto are rejected with an error.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace tdl {

struct Table;
struct Value {
    enum Type { NIL, BOOL, INT, FLT, STR, TAB, FUNC } type = NIL;
    bool b = false;
    long long i = 0;
    double d = 0.0;
    std::string s;
    std::shared_ptr<Table> tab;
    static Value nil() { return Value(); }
    static Value boolean(bool v) { Value x; x.type = BOOL; x.b = v; return x; }
    static Value integer(long long v) { Value x; x.type = INT; x.i = v; return x; }
    static Value number(double v) { Value x; x.type = FLT; x.d = v; return x; }
    static Value string(const std::string& v) { Value x; x.type = STR; x.s = v; return x; }
    bool truthy() const { return !(type == NIL || (type == BOOL && !b)); }
    bool is_number() const { return type == INT || type == FLT; }
    double as_double() const { return type == INT ? (double)i : d; }
};
struct Table {
    std::vector<Value> arr;              // t[1..n]
    std::map<std::string, Value> hash;   // named fields
};

struct LuaError {
    std::string msg;
};

// Host function: receives evaluated arguments, returns one value.  Throws LuaError on bad arguments.
using HostFn = std::function<Value(const std::vector<Value>&)>;

class Interp {
   public:
    Interp();
    void set_function(const std::string& name, HostFn fn);
    // Runs a chunk; returns false and fills err on a syntax or runtime error.
    bool run(const std::string& source, std::string* err);
    std::map<std::string, Value> globals;

   private:
    std::map<std::string, HostFn> fns_;
    friend struct Exec;
};

std::string tostring(const Value& v);

}  // namespace tdl
