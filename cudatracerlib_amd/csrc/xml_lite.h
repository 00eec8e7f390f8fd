// xml_lite.h — a small XML DOM reader (elements, attributes, comments, declarations, CDATA, the five predefined entities and
// numeric character references).  The reference parses scene files with pugixml (Engine/SceneLoader/Mitsuba/Utils.h:12), which
// is not vendored; the Mitsuba scene format needs nothing beyond this subset.
#pragma once
#include <string>
#include <vector>
#include <stdexcept>
#include <cctype>
#include <cstdlib>

namespace ctl {

struct xml_node {
    std::string name;                                             // as written; compare through lname()
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<xml_node> children;

    static std::string lower(std::string s) { for (auto& c : s) c = (char)std::tolower((unsigned char)c); return s; }
    std::string lname() const { return lower(name); }             // XMLNode::name() lower-cases (Utils.h:79-82)
    bool has_attr(const std::string& n) const { const std::string k = lower(n); for (auto& a : attrs) if (lower(a.first) == k) return true; return false; }
    const std::string& attr(const std::string& n) const {
        const std::string k = lower(n);
        for (auto& a : attrs) if (lower(a.first) == k) return a.second;
        throw std::runtime_error("no attributes in node!");       // XMLNode::get_attribute (Utils.h:108-114)
    }
    const xml_node* child(const std::string& n) const { const std::string k = lower(n); for (auto& c : children) if (c.lname() == k) return &c; return nullptr; }
    // Mitsuba "properties": child elements carrying name="..." (XMLNode::has_property / get_property, Utils.h:122-145)
    const xml_node* property(const std::string& n) const {
        const std::string k = lower(n);
        for (auto& c : children) if (c.has_attr("name") && lower(c.attr("name")) == k) return &c;
        return nullptr;
    }
};

class xml_parser {
    const std::string& s; size_t p = 0;
    int depth = 0;                                   // element() recurses once per nesting level: a scene file is a few levels deep, a file that nests deeper than
    static constexpr int kMaxDepth = 256;            // this is refused instead of running the parser (and every walk over the tree) out of stack
    [[noreturn]] void fail(const std::string& what) const {
        size_t line = 1; for (size_t i = 0; i < p && i < s.size(); i++) if (s[i] == '\n') line++;
        throw std::runtime_error("couldn't loader scene xml! (" + what + " at line " + std::to_string(line) + ")");
    }
    bool starts(const char* t) const { return s.compare(p, std::char_traits<char>::length(t), t) == 0; }
    void skip_ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) p++; }
    void skip_until(const char* t) { size_t e = s.find(t, p); if (e == std::string::npos) fail(std::string("unterminated ") + t); p = e + std::char_traits<char>::length(t); }
    static bool name_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
    std::string read_name() { size_t b = p; while (p < s.size() && name_char(s[p])) p++; if (b == p) fail("expected a name"); return s.substr(b, p - b); }
    static void append_utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    std::string unescape(const std::string& v) {
        std::string o; o.reserve(v.size());
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i] != '&') { o += v[i]; continue; }
            size_t e = v.find(';', i);
            if (e == std::string::npos) { o += v[i]; continue; }
            const std::string ent = v.substr(i + 1, e - i - 1);
            if (ent == "lt") o += '<'; else if (ent == "gt") o += '>'; else if (ent == "amp") o += '&'; else if (ent == "quot") o += '"'; else if (ent == "apos") o += '\'';
            else if (!ent.empty() && ent[0] == '#') append_utf8(o, (unsigned)std::strtoul(ent.c_str() + (ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X') ? 2 : 1), nullptr, ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X') ? 16 : 10));
            else { o += '&'; o += ent; o += ';'; }
            i = e;
        }
        return o;
    }
    void skip_misc() {
        for (;;) {
            skip_ws();
            if (starts("<!--")) { p += 4; skip_until("-->"); }
            else if (starts("<?")) { p += 2; skip_until("?>"); }
            else if (starts("<!DOCTYPE") || starts("<!doctype")) { skip_until(">"); }
            else return;
        }
    }
    xml_node element() {
        if (p >= s.size() || s[p] != '<') fail("expected '<'");
        p++;
        if (depth >= kMaxDepth) fail("elements nested more than " + std::to_string(kMaxDepth) + " deep");
        struct level { int& d; explicit level(int& x) : d(x) { d++; } ~level() { d--; } } guard(depth);
        xml_node n; n.name = read_name();
        for (;;) {
            skip_ws();
            if (p >= s.size()) fail("unterminated tag");
            if (s[p] == '/') { if (p + 1 >= s.size() || s[p + 1] != '>') fail("malformed empty-element tag"); p += 2; return n; }
            if (s[p] == '>') { p++; break; }
            std::string an = read_name();
            skip_ws(); if (p >= s.size() || s[p] != '=') fail("expected '=' after attribute name"); p++; skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) fail("expected a quoted attribute value");
            const char q = s[p++]; size_t e = s.find(q, p); if (e == std::string::npos) fail("unterminated attribute value");
            n.attrs.emplace_back(an, unescape(s.substr(p, e - p))); p = e + 1;
        }
        for (;;) {   // content
            size_t lt = s.find('<', p);
            if (lt == std::string::npos) fail("unterminated element <" + n.name + ">");
            p = lt;   // character data is not used by the scene format
            if (starts("<!--")) { p += 4; skip_until("-->"); }
            else if (starts("<![CDATA[")) { p += 9; skip_until("]]>"); }
            else if (starts("<?")) { p += 2; skip_until("?>"); }
            else if (starts("</")) {
                p += 2; std::string cn = read_name(); skip_ws();
                if (cn != n.name) fail("mismatched closing tag </" + cn + "> for <" + n.name + ">");
                if (p >= s.size() || s[p] != '>') fail("malformed closing tag"); p++;
                return n;
            }
            else n.children.push_back(element());
        }
    }
public:
    explicit xml_parser(const std::string& text) : s(text) {}
    // returns a synthetic document node whose children are the top-level elements (normally one)
    xml_node parse() {
        xml_node doc; doc.name = "";
        if (s.size() >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) p = 3;
        skip_misc();
        while (p < s.size()) { doc.children.push_back(element()); skip_misc(); }
        if (doc.children.empty()) fail("no root element");
        return doc;
    }
};

} // namespace ctl
