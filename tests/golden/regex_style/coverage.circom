// Test data: exercises the constructs of the loader's circom subset that the regex-style files do not
// (parameterised helper templates, recursion over a parameter, if / else on parameters, while, var arrays,
// integer functions with loops, ==> and -->, component arrays of parameterised templates, multi-dimensional
// signal arrays, array-valued anonymous outputs, nested anonymous calls).  The interpreter of oracle/circom
// executes the same text; values, names and the kept set must agree.
pragma circom 2.1.5;

include "./helpers.circom";

function log2ceil(n) {
    var r = 0;
    var v = n - 1;
    while (v > 0) { r++; v = v \ 2; }
    return r;
}

function weight(i) {
    if (i % 3 == 0) { return 1; } else if (i % 3 == 1) { return 2; }
    return 5;
}

// out[i] = in[i] AND sel  (an array-valued output of an anonymous component)
template GateAll(n) {
    signal input in[n];
    signal input sel;
    signal output out[n];
    for (var i = 0; i < n; i++) {
        out[i] <== in[i] * sel;
    }
}

// balanced OR tree, recursive in the parameter
template OrTree(n) {
    signal input in[n];
    signal output out;
    component l;
    component r;
    if (n == 1) {
        out <== in[0];
    } else if (n == 2) {
        out <== OR()(in[0], in[1]);
    } else {
        var h = n \ 2;
        l = OrTree(h);
        r = OrTree(n - h);
        for (var i = 0; i < h; i++) { l.in[i] <== in[i]; }
        for (var i = h; i < n; i++) { r.in[i - h] <== in[i]; }
        out <== l.out + r.out - l.out * r.out;
    }
}

// byte < bound, through a comparator whose width comes from a function
template Below(bound) {
    signal input in;
    signal output out;
    var bits = log2ceil(256);
    component lt = LessThan(bits);
    in ==> lt.in[0];
    lt.in[1] <== bound;
    lt.out ==> out;
}

template Coverage(msg_bytes) {
    signal input msg[msg_bytes];
    signal output out;
    signal output reveal0[msg_bytes];

    var thresholds[4] = [32, 65, 97, 128];
    component below[4][msg_bytes];
    signal cls[msg_bytes][4];
    for (var i = 0; i < msg_bytes; i++) {
        for (var k = 0; k < 4; k++) {
            below[k][i] = Below(thresholds[k]);
            below[k][i].in <== msg[i];
            cls[i][k] <== below[k][i].out;
        }
    }
    // weighted class sums (linear), their parity bit through a hint-based comparator, an OR tree over everything
    signal score[msg_bytes];
    signal odd[msg_bytes];
    component par[msg_bytes];
    var acc = 0;
    for (var i = 0; i < msg_bytes; i++) {
        var s = 0;
        for (var k = 0; k < 4; k++) { s += weight(k + i) * cls[i][k]; }
        score[i] <== s;
        par[i] = Num2Bits(5);
        par[i].in <== score[i];
        odd[i] <== par[i].out[0];
        acc += odd[i];
    }
    signal gated[msg_bytes];
    signal any <== OrTree(msg_bytes)(odd);
    gated <== GateAll(msg_bytes)(odd, any);
    var j = 0;
    signal run[msg_bytes + 1];
    run[0] <== 0;
    while (j < msg_bytes) {
        run[j + 1] <== XOR()(run[j], AND()(gated[j], NOT()(cls[j][0])));
        j += 1;
    }
    out <== IsEqual()([acc, run[msg_bytes] + acc - run[msg_bytes]]);
    for (var i = 0; i < msg_bytes; i++) {
        reveal0[i] <== msg[i] * run[i + 1];
    }
}
