// Known-answer tests of the per-block API (bwt_forward / bwt_inverse) on the GPU path.
// The ten vectors are the ones the reference's own unit test asserts
// (reference tests/test_small.cpp:80-92); each also checks length, presence of the
// delimiter and the round trip, as the reference does (:47-70).
#include <iostream>
#include <string>
#include <vector>

#include "bwt.hpp"
#include "inverse_bwt.hpp"

struct Case {
    const char* name;
    std::string input;
    char delimiter;
    std::string expected;  // "" = only round trip
    bool misuse = false;   // delimiter occurs in the input: the inverse stops early, as the reference's does
    std::string expected_back = "";
};

static bool check(const Case& c, std::string& why) {
    const std::string fwd = bwt_forward(c.input, c.delimiter);
    if (fwd.size() != c.input.size() + 1) {
        why = "forward length " + std::to_string(fwd.size()) + ", expected " + std::to_string(c.input.size() + 1);
        return false;
    }
    if (fwd.find(c.delimiter) == std::string::npos) {
        why = "delimiter missing from the forward result";
        return false;
    }
    if (!c.expected.empty() && fwd != c.expected) {
        why = "forward mismatch: expected \"" + c.expected + "\", got \"" + fwd + "\"";
        return false;
    }
    const std::string back = bwt_inverse(fwd, c.delimiter);
    if (back != (c.misuse ? c.expected_back : c.input)) {
        why = "round trip mismatch: got \"" + back + "\"";
        return false;
    }
    return true;
}

int main() {
    std::cout << "Running BWT tests...\n" << std::endl;
    const std::vector<Case> cases = {
        {"Forward BWT: basic string (with known result)", "banana", '~', "bnn~aaa"},
        {"Forward BWT: hello", "hello", '~', "h~ello"},
        {"Forward BWT: mississippi", "mississippi", '~', "ssmp~pissiii"},
        {"Forward BWT: empty string", "", '~', "~"},
        {"Forward BWT: single character", "a", '~', "~a"},
        {"Forward BWT: repeated characters", "aaaa", '~', "~aaaa"},
        {"Forward BWT: special characters", "a!b@c#", '~', "acb~!@#"},
        {"Forward BWT: custom delimiter", "test", '$', "ttes$"},
        {"Forward BWT: longer string", "the quick brown fox jumps over the lazy dog", '~',
         "kynxeserl i hhv ottu c uwd rfm ebp ~qjooozag"},
        {"Forward BWT: string with newlines", "line1\nline2\nline3", '~', "12eeennnlll~\n\niii3"},
        // beyond the reference's list: API misuse and unsigned ordering (expected bytes from the reference)
        {"Forward BWT: delimiter inside the input", "ab~ab", '~', "~~aabb", true, "ab"},
        {"Forward BWT: high bytes sort unsigned", std::string("a\0b\xff" "a", 5), '\x01', std::string("aa\x01\xff\0b", 6)},
    };
    int passed = 0, failed = 0;
    for (const Case& c : cases) {
        std::string why;
        const bool ok = check(c, why);
        std::cout << "[" << (ok ? "PASS" : "FAIL") << "] " << c.name;
        if (!ok) std::cout << " - " << why;
        std::cout << std::endl;
        (ok ? passed : failed)++;
    }
    std::cout << "\n=== Test Summary ===\nTotal tests: " << cases.size() << "\nPassed: " << passed
              << "\nFailed: " << failed << std::endl;
    std::cout << (failed ? "\nSome tests failed!" : "\nAll tests passed!") << std::endl;
    return failed ? 1 : 0;
}
