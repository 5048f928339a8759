// Json.h -- a small JSON DOM (RFC 8259) for the glTF importer: objects, arrays, numbers (double), strings with
// escapes (\uXXXX -> UTF-8), booleans, null.  Throws PathTracing::error on malformed input.
#pragma once

#include <cstdlib>
#include <map>
#include <string>
#include <string_view>
#include <vector>

#include "Scene.h"

namespace PathTracing
{

class Json
{
public:
    enum class Kind { Null, Bool, Number, String, Array, Object };

    Kind kind = Kind::Null;
    bool boolean = false;
    double number = 0.0;
    std::string string;
    std::vector<Json> array;
    std::map<std::string, Json> object;

    static Json Parse(std::string_view text)
    {
        size_t pos = 0;
        Json v = ParseValue(text, pos, 0);
        SkipSpace(text, pos);
        if (pos != text.size())
            throw error("JSON: trailing characters");
        return v;
    }

    [[nodiscard]] bool Has(const std::string &key) const { return kind == Kind::Object && object.count(key) != 0; }
    // missing keys read as null, so chains like j["a"]["b"].Num(1.0) are safe
    const Json &operator[](const std::string &key) const
    {
        static const Json null;
        if (kind != Kind::Object)
            return null;
        const auto it = object.find(key);
        return it == object.end() ? null : it->second;
    }
    const Json &operator[](size_t i) const
    {
        static const Json null;
        return (kind == Kind::Array && i < array.size()) ? array[i] : null;
    }
    [[nodiscard]] size_t Size() const { return kind == Kind::Array ? array.size() : 0; }
    [[nodiscard]] bool IsNull() const { return kind == Kind::Null; }
    [[nodiscard]] double Num(double fallback = 0.0) const { return kind == Kind::Number ? number : fallback; }
    [[nodiscard]] int64_t Int(int64_t fallback = -1) const { return kind == Kind::Number ? static_cast<int64_t>(number) : fallback; }
    [[nodiscard]] const std::string &Str() const { return string; }

private:
    static void SkipSpace(std::string_view t, size_t &p)
    {
        while (p < t.size() && (t[p] == ' ' || t[p] == '\t' || t[p] == '\n' || t[p] == '\r'))
            p++;
    }
    static Json ParseValue(std::string_view t, size_t &p, int depth)
    {
        if (depth > 256)
            throw error("JSON: nesting too deep");
        SkipSpace(t, p);
        if (p >= t.size())
            throw error("JSON: unexpected end");
        Json v;
        const char c = t[p];
        if (c == '{')
        {
            v.kind = Kind::Object;
            p++;
            SkipSpace(t, p);
            if (p < t.size() && t[p] == '}') { p++; return v; }
            for (;;)
            {
                SkipSpace(t, p);
                if (p >= t.size() || t[p] != '"') throw error("JSON: expected a key");
                std::string key = ParseString(t, p);
                SkipSpace(t, p);
                if (p >= t.size() || t[p] != ':') throw error("JSON: expected ':'");
                p++;
                v.object[std::move(key)] = ParseValue(t, p, depth + 1);
                SkipSpace(t, p);
                if (p < t.size() && t[p] == ',') { p++; continue; }
                if (p < t.size() && t[p] == '}') { p++; return v; }
                throw error("JSON: expected ',' or '}'");
            }
        }
        if (c == '[')
        {
            v.kind = Kind::Array;
            p++;
            SkipSpace(t, p);
            if (p < t.size() && t[p] == ']') { p++; return v; }
            for (;;)
            {
                v.array.push_back(ParseValue(t, p, depth + 1));
                SkipSpace(t, p);
                if (p < t.size() && t[p] == ',') { p++; continue; }
                if (p < t.size() && t[p] == ']') { p++; return v; }
                throw error("JSON: expected ',' or ']'");
            }
        }
        if (c == '"')
        {
            v.kind = Kind::String;
            v.string = ParseString(t, p);
            return v;
        }
        if (t.compare(p, 4, "true") == 0) { v.kind = Kind::Bool; v.boolean = true; p += 4; return v; }
        if (t.compare(p, 5, "false") == 0) { v.kind = Kind::Bool; p += 5; return v; }
        if (t.compare(p, 4, "null") == 0) { p += 4; return v; }
        const size_t start = p;
        while (p < t.size() && (std::isdigit(static_cast<unsigned char>(t[p])) || t[p] == '-' || t[p] == '+' || t[p] == '.' || t[p] == 'e' || t[p] == 'E'))
            p++;
        if (p == start)
            throw error("JSON: unexpected character");
        v.kind = Kind::Number;
        v.number = std::strtod(std::string(t.substr(start, p - start)).c_str(), nullptr);
        return v;
    }
    static std::string ParseString(std::string_view t, size_t &p)
    {
        std::string s;
        p++; // opening quote
        while (p < t.size() && t[p] != '"')
        {
            char c = t[p++];
            if (c != '\\')
            {
                s.push_back(c);
                continue;
            }
            if (p >= t.size()) break;
            c = t[p++];
            switch (c)
            {
            case 'n': s.push_back('\n'); break;
            case 't': s.push_back('\t'); break;
            case 'r': s.push_back('\r'); break;
            case 'b': s.push_back('\b'); break;
            case 'f': s.push_back('\f'); break;
            case 'u':
            {
                if (p + 4 > t.size()) throw error("JSON: bad \\u escape");
                const uint32_t cp = static_cast<uint32_t>(std::strtoul(std::string(t.substr(p, 4)).c_str(), nullptr, 16));
                p += 4;
                if (cp < 0x80) s.push_back(static_cast<char>(cp));
                else if (cp < 0x800) { s.push_back(static_cast<char>(0xc0 | (cp >> 6))); s.push_back(static_cast<char>(0x80 | (cp & 63))); }
                else { s.push_back(static_cast<char>(0xe0 | (cp >> 12))); s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 63))); s.push_back(static_cast<char>(0x80 | (cp & 63))); }
                break;
            }
            default: s.push_back(c); break; // \" \\ \/
            }
        }
        if (p >= t.size())
            throw error("JSON: unterminated string");
        p++; // closing quote
        return s;
    }
};

}
