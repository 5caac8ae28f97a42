"""CPU check of the arithmetic behind the opt-in split-bf16 projection GEMM (climsim_amd/csrc/gemm.hip::proj_gemm_b3_kernel,
DESIGN.md 4.10), restated in numpy: (1) an fp32 value is EXACTLY the sum of three bf16 values obtained by rounding to nearest
three times; (2) every bf16 x bf16 partial product is exact in fp32; (3) the three dropped cross terms are at most 2^-24 |a||b|
each; (4) accumulating the six kept products in fp32 lands as close to the float64 dot product as an fp32 multiply-add chain.
The GPU test (tests/test_gemm_split.py) measures the same on the kernel itself."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo


def _values(rng, n):
    x = (rng.standard_normal(n) * np.exp(6.0 * rng.standard_normal(n))).astype(np.float32)
    x[:8] = [1.0, -1.0, 1.0 + 2.0 ** -23, 3.0e38, -3.0e38, 1.5e-30, 0.0, np.float32(np.pi)]
    return x


def test_three_way_bf16_split_of_an_fp32_value_is_exact():
    x = _values(np.random.Generator(np.random.PCG64(1)), 200_000)
    hi, mid, lo = split3(x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= 2.0 ** -8 * np.abs(x[nz]) * (1 + 2.0 ** -7))          # half an ulp of bf16
    assert np.all(np.abs(lo[nz]) <= 2.0 ** -16 * np.abs(x[nz]) * (1 + 2.0 ** -6))


def test_partial_products_are_exact_in_fp32_and_dropped_terms_are_below_one_rounding():
    rng = np.random.Generator(np.random.PCG64(2))
    a, b = _values(rng, 100_000), _values(rng, 100_000)
    keep = np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e37           # away from overflow of the product itself
    a, b = a[keep], b[keep]
    A, B = split3(a), split3(b)
    for p in A:
        for q in B:
            exact = p.astype(np.float64) * q.astype(np.float64)
            ok = np.abs(exact) > 1e-35                                           # above the subnormal range of fp32
            assert np.array_equal((p * q).astype(np.float64)[ok], exact[ok])
    ab = np.abs(a.astype(np.float64) * b.astype(np.float64))
    for i, j in ((1, 2), (2, 1), (2, 2)):                                        # mid*lo, lo*mid, lo*lo
        d = np.abs(A[i].astype(np.float64) * B[j].astype(np.float64))
        assert np.all(d <= 2.0 ** -24 * ab * (1 + 2.0 ** -5))


def test_six_product_fp32_accumulation_is_as_close_to_float64_as_an_fp32_chain():
    rng = np.random.Generator(np.random.PCG64(3))
    M, K = 4096, 144
    X = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, 1)))).astype(np.float32)
    w = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    ref = X.astype(np.float64) @ w.astype(np.float64)
    # fp32 chain: acc = fl(acc + fl-exact(a*b)) with a fused multiply-add per k (float64 product rounded once to fp32 = fma)
    chain = np.zeros(M, np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + X[:, k].astype(np.float64) * np.float64(w[k])).astype(np.float32)
    # split: per 16-deep block the six kept products, smallest first, each added to the fp32 accumulator
    XS, WS = split3(X), split3(w)
    acc = np.zeros(M, np.float32)
    order = ((1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0))
    for k0 in range(0, K, 16):
        for i, j in order:
            # one MFMA: the 16 exact products of the block summed (wider internal sum), then one rounding into the accumulator
            blk = (XS[i][:, k0:k0 + 16].astype(np.float64) * WS[j][k0:k0 + 16].astype(np.float64)).sum(1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    mag = np.abs(X.astype(np.float64)) @ np.abs(w.astype(np.float64))
    e_chain, e_split = np.abs(chain - ref), np.abs(acc - ref)
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    print(f"max err / sum|a||b|: chain {np.max(e_chain / mag):.2e} split {np.max(e_split / mag):.2e}; rms chain {rms(e_chain):.3e} split {rms(e_split):.3e}")
    assert np.max(e_split / mag) < 1e-6
    assert rms(e_split) <= 1.25 * rms(e_chain)


def test_operand_image_of_the_split_gemm_is_bank_conflict_free():
    """The kernel's LDS image (rows of 16 bf16 = 32 B, the two 16-byte halves of rows 8..15 mod 16 swapped) under gfx950's banking
    rules (/opt/skills/guides/MI355X_MICROARCH.md, LDS table): ds_read_b128 = four 16-lane groups {0-3,12-15,20-27},
    {4-11,16-19,28-31} (+32), bank (a/4) mod 64; ds_write_b64 = four contiguous 16-lane groups, bank (a/4) mod 32."""
    LD = 16
    swz = lambda r: (r >> 3) & 1
    g0, g1 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    read_groups = [g0, g1, [l + 32 for l in g0], [l + 32 for l in g1]]

    def worst(groups, addr, nbytes, nbanks):
        w = 0
        for g in groups:
            use = {}
            for l in g:
                for d in range(nbytes // 4):
                    use.setdefault((addr(l) // 4 + d) % nbanks, set()).add(addr(l) // 4 + d)
            w = max(w, max(len(v) for v in use.values()))
        return w

    # operand read of lane l: row l & 31 of the wave's 32-row tile, k-group l >> 5 (gemm.hip: arow / wrow)
    assert worst(read_groups, lambda l: 2 * ((l & 31) * LD + 8 * ((l >> 5) ^ swz(l & 31))), 16, 64) == 1
    assert worst(read_groups, lambda l: 2 * ((l & 31) * LD + 8 * (l >> 5)), 16, 64) == 2           # without the swap: two-way
    # staging store of thread t of a wave: row t >> 2, four bf16 at k = 4 (t & 3) (gemm.hip: o0)
    write_groups = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
    waddr = lambda t: 2 * ((t >> 2) * LD + 8 * ((((t & 3) * 4) >> 3) ^ swz(t >> 2)) + (((t & 3) * 4) & 7))
    assert worst(write_groups, waddr, 8, 32) == 1
