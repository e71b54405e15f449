// eg_json.h — a minimal JSON reader (objects keep insertion order) for checkpoints and the CLI's data files.
#pragma once
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace eg {

struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false; double num = 0.0; std::string str;
  std::vector<Json> arr; std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const char* key) const { for (auto& kv : obj) if (kv.first == key) return &kv.second; return nullptr; }
};
struct JsonParser {
  const char* p; const char* end; std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool fail(const char* m) { if (err.empty()) err = m; return false; }
  bool string(std::string& s) {
    if (p >= end || *p != '"') return fail("expected string");
    ++p; s.clear();
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) return fail("bad escape");
        switch (*p) { case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break; case 'b': s += '\b'; break;
                      case 'f': s += '\f'; break; case 'u': { if (end - p < 5) return fail("bad \\u"); unsigned c = std::strtoul(std::string(p + 1, 4).c_str(), nullptr, 16); s += char(c < 128 ? c : '?'); p += 4; break; }
                      default: s += *p; }
        ++p;
      } else s += *p++;
    }
    if (p >= end) return fail("unterminated string");
    ++p; return true;
  }
  bool value(Json& j, int depth = 0) {
    if (depth > 64) return fail("nesting too deep");
    ws(); if (p >= end) return fail("unexpected end");
    if (*p == '{') {
      j.kind = Json::Obj; ++p; ws();
      if (p < end && *p == '}') { ++p; return true; }
      while (true) {
        ws(); std::string k; if (!string(k)) return false;
        ws(); if (p >= end || *p != ':') return fail("expected ':'"); ++p;
        j.obj.emplace_back(k, Json()); if (!value(j.obj.back().second, depth + 1)) return false;
        ws(); if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      j.kind = Json::Arr; ++p; ws();
      if (p < end && *p == ']') { ++p; return true; }
      while (true) {
        j.arr.emplace_back(); if (!value(j.arr.back(), depth + 1)) return false;
        ws(); if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { j.kind = Json::Str; return string(j.str); }
    if (end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; j.kind = Json::Null; return true; }
    if (end - p >= 4 && !std::strncmp(p, "true", 4)) { p += 4; j.kind = Json::Bool; j.b = true; return true; }
    if (end - p >= 5 && !std::strncmp(p, "false", 5)) { p += 5; j.kind = Json::Bool; j.b = false; return true; }
    char* e = nullptr; errno = 0; double v = std::strtod(p, &e);
    if (e == p) return fail("bad token");
    j.kind = Json::Num; j.num = v; p = e; return true;
  }
};


}  // namespace eg
