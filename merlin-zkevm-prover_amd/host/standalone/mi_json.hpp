// mi_json.hpp -- a small JSON reader for the host layer where nlohmann/json is not available (this image; the standalone tools).
// Numbers are kept as text and read on demand as u64 (starkinfo offsets exceed 2^53: no doubles), objects keep their key order.
#ifndef MI_JSON_HPP
#define MI_JSON_HPP
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mi {
class Json
{
public:
    enum Kind { Null, Bool, Number, String, Array, Object };
    Kind kind = Null;
    std::string text;                                  // Number (as written), String (unescaped), Bool ("true" / "false")
    std::vector<Json> items;                           // Array
    std::vector<std::pair<std::string, Json>> members; // Object, in file order

    bool isNull() const { return kind == Null; }
    size_t size() const { return kind == Array ? items.size() : kind == Object ? members.size() : 0; }
    bool contains(const std::string &k) const { return find(k) != nullptr; }
    const Json *find(const std::string &k) const
    {
        if (kind == Object)
            for (const auto &m : members) if (m.first == k) return &m.second;
        return nullptr;
    }
    const Json &operator[](const std::string &k) const
    {
        const Json *j = find(k);
        if (!j) throw std::runtime_error("json: key \"" + k + "\" missing");
        return *j;
    }
    const Json &operator[](size_t i) const
    {
        if (kind != Array || i >= items.size()) throw std::runtime_error("json: index out of range");
        return items[i];
    }
    uint64_t u64() const
    {
        if (kind != Number && kind != String) throw std::runtime_error("json: not a number");
        return std::strtoull(text.c_str(), nullptr, 10);
    }
    bool boolean() const
    {
        if (kind != Bool) throw std::runtime_error("json: not a boolean");
        return text == "true";
    }
    const std::string &str() const
    {
        if (kind != String) throw std::runtime_error("json: not a string");
        return text;
    }

    static Json parse(const std::string &s)
    {
        size_t p = 0;
        Json j = value(s, p);
        ws(s, p);
        if (p != s.size()) throw std::runtime_error("json: trailing characters at offset " + std::to_string(p));
        return j;
    }
    static Json parseFile(const std::string &path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("json: cannot open " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        return parse(ss.str());
    }

private:
    static void ws(const std::string &s, size_t &p)
    {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++;
    }
    [[noreturn]] static void bad(size_t p, const char *what) { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p)); }
    static std::string quoted(const std::string &s, size_t &p)
    {
        std::string out;
        for (p++; p < s.size() && s[p] != '"'; p++) {
            if (s[p] != '\\') { out += s[p]; continue; }
            if (++p >= s.size()) break;
            switch (s[p]) {
            case 'n': out += '\n'; break;
            case 't': out += '\t'; break;
            case 'r': out += '\r'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'u': // \uXXXX: only the ASCII range occurs in the files this reads
                if (p + 4 < s.size()) { out += (char)std::strtoul(s.substr(p + 1, 4).c_str(), nullptr, 16); p += 4; }
                break;
            default: out += s[p];
            }
        }
        if (p >= s.size()) bad(p, "unterminated string");
        p++;
        return out;
    }
    static Json value(const std::string &s, size_t &p)
    {
        ws(s, p);
        if (p >= s.size()) bad(p, "unexpected end");
        Json j;
        const char ch = s[p];
        if (ch == '{') {
            j.kind = Object;
            p++;
            ws(s, p);
            if (p < s.size() && s[p] == '}') { p++; return j; }
            for (;;) {
                ws(s, p);
                if (p >= s.size() || s[p] != '"') bad(p, "expected a key");
                std::string k = quoted(s, p);
                ws(s, p);
                if (p >= s.size() || s[p] != ':') bad(p, "expected ':'");
                p++;
                j.members.emplace_back(std::move(k), value(s, p));
                ws(s, p);
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == '}') { p++; return j; }
                bad(p, "expected ',' or '}'");
            }
        }
        if (ch == '[') {
            j.kind = Array;
            p++;
            ws(s, p);
            if (p < s.size() && s[p] == ']') { p++; return j; }
            for (;;) {
                j.items.push_back(value(s, p));
                ws(s, p);
                if (p < s.size() && s[p] == ',') { p++; continue; }
                if (p < s.size() && s[p] == ']') { p++; return j; }
                bad(p, "expected ',' or ']'");
            }
        }
        if (ch == '"') { j.kind = String; j.text = quoted(s, p); return j; }
        if (s.compare(p, 4, "true") == 0) { j.kind = Bool; j.text = "true"; p += 4; return j; }
        if (s.compare(p, 5, "false") == 0) { j.kind = Bool; j.text = "false"; p += 5; return j; }
        if (s.compare(p, 4, "null") == 0) { p += 4; return j; }
        const size_t b = p;
        while (p < s.size() && (std::isdigit((unsigned char)s[p]) || s[p] == '-' || s[p] == '+' || s[p] == '.' || s[p] == 'e' || s[p] == 'E')) p++;
        if (p == b) bad(p, "unexpected character");
        j.kind = Number;
        j.text = s.substr(b, p - b);
        return j;
    }
};
} // namespace mi
#endif
