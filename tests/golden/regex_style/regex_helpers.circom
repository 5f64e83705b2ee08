// zkwg stand-in for @zk-email/zk-regex-circom/circuits/regex_helpers.circom (see
// tools/gen_body_hash_regex.py; the real file is absent offline).
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
