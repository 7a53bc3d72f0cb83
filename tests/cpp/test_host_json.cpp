// CPU-only check of the starkstruct parsing in host/build_const_tree.hpp: top-level keys only, whatever their order.
#include <cstdio>
#include "build_const_tree.hpp"
int main()
{
    using bctree_detail::json_field;
    std::string v;
    const std::string a = "{\n \"steps\": [ {\"nBits\": 24}, {\"nBits\": 19} ],\n \"nBits\": 23, \"nBitsExt\":24,\n\"verificationHashType\" : \"GL\", \"nQueries\": 128 }";
    if (!json_field(a, "nBits", v) || v != "23") { std::printf("nBits -> %s\n", v.c_str()); return 1; }
    if (!json_field(a, "nBitsExt", v) || v != "24") return 2;
    if (!json_field(a, "verificationHashType", v) || v != "GL") return 3;
    if (!json_field(a, "nQueries", v) || v != "128") return 4;
    if (json_field(a, "missing", v)) return 5;
    const std::string b = "{\"x\": {\"nBits\": 1}, \"s\": \"a \\\"nBits\\\": 7 b\", \"nBits\":9}";
    if (!json_field(b, "nBits", v) || v != "9") { std::printf("nested/string -> %s\n", v.c_str()); return 6; }
    const std::string c = "{\"steps\": [{\"nBits\": 5}]}";
    if (json_field(c, "nBits", v)) return 7; // only nested occurrences: not found
    std::printf("json ok\n");
    return 0;
}
