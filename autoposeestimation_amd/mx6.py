"""The "F16M6" operand line of csrc/conv3x3_halo_mx.hip: 32 consecutive channels of one pixel (or of one weight row and tap) in
128 bytes, so that the LDS-DMA / swizzle machinery of the pre-split ("S32") kernels carries it unchanged:

    bytes   0 ..  63   v1 = fp16(v), 32 values                                  the main product's operand  (v_mfma_f32_16x16x32_f16)
    bytes  64 ..  79   codes 0 .. 127 (bits) of Q(v1): block-scaled e2m3, 6 bits each, value i in bits [6 i, 6 i + 5]
    bytes  80 ..  95   the same of Q(v2), the residual v2 = v - v1
    bytes  96 .. 103   bits 128 .. 191 of Q(v1);  byte 104 its E8M0 scale exponent (2^(E - 127));  bytes 105 .. 111 zero
    bytes 112 .. 119   bits 128 .. 191 of Q(v2);  byte 120 its scale exponent;  bytes 121 .. 127 zero
    (the two fields' 16-byte chunks alternate -- 4, 6 = Q(v1), 5, 7 = Q(v2) -- so that the lane groups of one ds_read_b128, which read
    Q(v1) on even and Q(v2) on odd 16-lane quarters, land on different LDS slots under the kernels' XOR swizzle)

    x . w  ~  x1 w1  +  Q(x1) Q(w2)  +  Q(x2) Q(w1)          (the two cross terms through v_mfma_scale_f32_16x16x128_f8f6f4, 6 of 12 passes)

Block scale: 2^(floor(log2 max|v|) - 2), i.e. the block maximum lands in [4, 8) of e2m3's range (largest code 7.5); codes are rounded to
nearest-even on e2m3's grid (subnormal step 1/8 below 1).  This module is the host-side packer (weights, once per model) and the reference
the tests decode the device converter's output with."""
import numpy as np

_E2M3_VALUES = np.array([(m / 8.0 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 1)) for e in range(4) for m in range(8)], dtype=np.float32)


def e2m3_decode(codes):
    """uint8 codes (sign << 5 | exponent << 3 | mantissa) -> float32"""
    codes = np.asarray(codes)
    v = _E2M3_VALUES[codes & 31]
    return np.where(codes & 32, -v, v).astype(np.float32)


def quantise_blocks(v):
    """v[..., 32] float32 -> (codes uint8[..., 32], e8m0 uint8[...]); v ~ e2m3_decode(codes) * 2^(e8m0 - 127)"""
    v = np.asarray(v, dtype=np.float32)
    amax = np.abs(v).max(axis=-1)
    bits = amax.view(np.uint32)
    ex = ((bits >> 23) & 255).astype(np.int32) - 127                    # floor(log2 amax) of a normal float
    es = np.clip(ex - 2, -127, 127)
    es = np.where(amax == 0, 0, es)
    u = v / np.exp2(es.astype(np.float32))[..., None]
    a = np.minimum(np.abs(u), np.float32(7.5))
    sub = a < 1.0
    m_sub = np.rint(a * 8.0).astype(np.int32)                            # 0 .. 8 (8 = the first normal code)
    exn = np.floor(np.log2(np.maximum(a, 1.0))).astype(np.int32)         # 0, 1, 2
    q = np.rint(a / np.exp2((exn - 3).astype(np.float32))).astype(np.int32)      # 8 .. 16
    carry = q == 16
    exn = np.where(carry, exn + 1, exn)
    q = np.where(carry, 8, q)
    over = exn > 2                                                       # (7.5 < a rounded up past the grid: not reachable after the clamp)
    code_n = np.where(over, (3 << 3) | 7, ((exn + 1) << 3) | (q - 8))
    code_s = np.where(m_sub == 8, 1 << 3, m_sub)
    code = np.where(sub, code_s, code_n).astype(np.uint8)
    code |= (np.signbit(u) & (code != 0)).astype(np.uint8) << 5
    return code, (es + 127).astype(np.uint8)


def _pack6(codes):
    """uint8[..., 32] six-bit codes -> uint8[..., 24] (little-endian bit stream)"""
    c = codes.astype(np.uint64)
    out = np.zeros(codes.shape[:-1] + (3,), dtype=np.uint64)
    for i in range(32):
        bit = 6 * i
        w, s = bit // 64, bit % 64
        out[..., w] |= (c[..., i] << np.uint64(s)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        if s > 58:
            out[..., w + 1] |= c[..., i] >> np.uint64(64 - s)
    return out.view(np.uint8).reshape(codes.shape[:-1] + (24,))


def _unpack6(b):
    """uint8[..., 24] -> uint8[..., 32]"""
    w = np.ascontiguousarray(b).view(np.uint64).reshape(b.shape[:-1] + (3,))
    out = np.zeros(b.shape[:-1] + (32,), dtype=np.uint8)
    for i in range(32):
        bit = 6 * i
        k, s = bit // 64, bit % 64
        v = w[..., k] >> np.uint64(s)
        if s > 58:
            v = v | (w[..., k + 1] << np.uint64(64 - s))
        out[..., i] = (v & np.uint64(63)).astype(np.uint8)
    return out


def pack_lines(v):
    """v[..., 32] float32 -> uint8[..., 128] F16M6 lines"""
    v = np.asarray(v, dtype=np.float32)
    v1 = v.astype(np.float16)
    v2 = v - v1.astype(np.float32)
    c1, s1 = quantise_blocks(v1.astype(np.float32))
    c2, s2 = quantise_blocks(v2)
    line = np.zeros(v.shape[:-1] + (128,), dtype=np.uint8)
    line[..., 0:64] = v1.view(np.uint8).reshape(v.shape[:-1] + (64,))
    p1, p2 = _pack6(c1), _pack6(c2)
    line[..., 64:80], line[..., 96:104], line[..., 104] = p1[..., :16], p1[..., 16:], s1
    line[..., 80:96], line[..., 112:120], line[..., 120] = p2[..., :16], p2[..., 16:], s2
    return line


def unpack_lines(line):
    """uint8[..., 128] -> (v1 float32[..., 32], Q(v1) float32[..., 32], Q(v2) float32[..., 32]): the three operands a line carries"""
    line = np.ascontiguousarray(line)
    v1 = line[..., 0:64].copy().view(np.float16).reshape(line.shape[:-1] + (32,)).astype(np.float32)
    p1 = np.concatenate([line[..., 64:80], line[..., 96:104]], axis=-1)
    p2 = np.concatenate([line[..., 80:96], line[..., 112:120]], axis=-1)
    q1 = e2m3_decode(_unpack6(p1)) * np.exp2(line[..., 104].astype(np.float32) - 127.0)[..., None]
    q2 = e2m3_decode(_unpack6(p2)) * np.exp2(line[..., 120].astype(np.float32) - 127.0)[..., None]
    return v1, q1.astype(np.float32), q2.astype(np.float32)


def pack_conv_weights(w):
    """conv weight [Cout, Cin, 3, 3] (Cin % 32 == 0) -> uint8[Cout, 9 * Cin / 32, 128]: k-group = tap * (Cin / 32) + chunk, as the S32K layout"""
    w = np.asarray(w, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    if cin % 32:
        raise ValueError("Cin % 32")
    v = np.transpose(w, (0, 2, 3, 1)).reshape(cout, kh * kw * (cin // 32), 32)
    return pack_lines(v)
