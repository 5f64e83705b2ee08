// Helper templates for tests/golden/regex_style/*.circom (test data, written for this repository in the
// style of zk-regex's regex_helpers.circom, which is not available offline).
pragma circom 2.1.5;

include "circomlib/circuits/comparators.circom";
include "circomlib/circuits/gates.circom";

template MultiOR(n) {
    signal input in[n];
    signal output out;

    signal sums[n];
    sums[0] <== in[0];
    for (var i = 1; i < n; i++) {
        sums[i] <== sums[i-1] + in[i];
    }

    component is_zero = IsZero();
    is_zero.in <== sums[n-1];
    out <== 1 - is_zero.out;
}

template MultiNOR(n) {
    signal input in[n];
    signal output out;

    var total = 0;
    for (var i = 0; i < n; i++) {
        total += in[i];
    }

    component is_zero = IsZero();
    is_zero.in <== total;
    out <== is_zero.out;
}

// out = in[0] | (in[1] & in[2])
template ORAnd() {
    signal input in[3];
    signal output out;

    signal and_out <== in[1] * in[2];
    out <== in[0] + and_out - in[0] * and_out;
}
