// bt_json.hpp -- minimal JSON document parser for scene.json[.gz] (replaces serde_json in
// the reference's main.rs:93-102).  Recursive descent; numbers keep their source text so
// f32 fields are converted with strtof (correctly rounded from the decimal, as serde does)
// and u64 keys with strtoull.  Handles -0.0, exponents (1.8626451e-09), null, escapes.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace btjson {

struct Value;
using ValuePtr = std::unique_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string text;                                       // Number (source text) or String
    std::vector<ValuePtr> items;                            // Array
    std::vector<std::pair<std::string, ValuePtr>> members;  // Object, in file order

    bool is_null() const { return kind == Null; }
    const Value *find(const std::string &key) const {
        if (kind != Object) return nullptr;
        for (auto &m : members)
            if (m.first == key) return m.second.get();
        return nullptr;
    }
    const Value &at(const std::string &key) const {
        const Value *v = find(key);
        if (!v) throw std::runtime_error("missing field `" + key + "`");
        return *v;
    }
    const Value &at(size_t i) const {
        if (kind != Array || i >= items.size()) throw std::runtime_error("array index out of range");
        return *items[i];
    }
    float as_f32() const {
        if (kind != Number) throw std::runtime_error("expected a number");
        return std::strtof(text.c_str(), nullptr);
    }
    uint64_t as_u64() const {
        if (kind != Number) throw std::runtime_error("expected an unsigned integer");
        if (text.find_first_of(".eE-") != std::string::npos) throw std::runtime_error("expected an unsigned integer");
        return std::strtoull(text.c_str(), nullptr, 10);
    }
    const std::string &as_string() const {
        if (kind != String) throw std::runtime_error("expected a string");
        return text;
    }
};

class Parser {
  public:
    Parser(const char *s, size_t n) : p_(s), end_(s + n) {}
    ValuePtr parse() {
        ValuePtr v = value(0);
        ws();
        if (p_ != end_) fail("trailing characters");
        return v;
    }

  private:
    const char *p_, *end_;
    [[noreturn]] void fail(const char *msg) { throw std::runtime_error(std::string("JSON: ") + msg); }
    void ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    bool lit(const char *w) {
        size_t n = std::char_traits<char>::length(w);
        if ((size_t)(end_ - p_) >= n && std::char_traits<char>::compare(p_, w, n) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }
    ValuePtr value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p_ == end_) fail("unexpected end of input");
        ValuePtr v(new Value());
        char c = *p_;
        if (c == '{') {
            ++p_;
            v->kind = Value::Object;
            ws();
            if (p_ < end_ && *p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                if (p_ == end_ || *p_ != '"') fail("expected a member name");
                std::string key = string();
                ws();
                if (p_ == end_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v->members.emplace_back(std::move(key), value(depth + 1));
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            ++p_;
            v->kind = Value::Array;
            ws();
            if (p_ < end_ && *p_ == ']') { ++p_; return v; }
            for (;;) {
                v->items.push_back(value(depth + 1));
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = Value::String;
            v->text = string();
        } else if (lit("null")) {
            v->kind = Value::Null;
        } else if (lit("true")) {
            v->kind = Value::Bool;
            v->b = true;
        } else if (lit("false")) {
            v->kind = Value::Bool;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char *s = p_;
            if (*p_ == '-') ++p_;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("malformed number");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
            if (p_ < end_ && *p_ == '.') {
                ++p_;
                if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("malformed number");
                while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
            }
            if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
                ++p_;
                if (p_ < end_ && (*p_ == '+' || *p_ == '-')) ++p_;
                if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("malformed number");
                while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
            }
            v->kind = Value::Number;
            v->text.assign(s, p_);
        } else {
            fail("unexpected character");
        }
        return v;
    }
    static void utf8(std::string &out, uint32_t cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
        }
    }
    uint32_t hex4() {
        if (end_ - p_ < 4) fail("bad \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        ++p_; // opening quote
        std::string out;
        while (p_ < end_ && *p_ != '"') {
            char c = *p_++;
            if ((unsigned char)c < 0x20) fail("control character in string");
            if (c != '\\') { out += c; continue; }
            if (p_ == end_) fail("bad escape");
            char e = *p_++;
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                uint32_t cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                    p_ += 2;
                    uint32_t lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
        if (p_ == end_) fail("unterminated string");
        ++p_;
        return out;
    }
};

inline ValuePtr parse(const char *s, size_t n) { return Parser(s, n).parse(); }

} // namespace btjson
