"""Host-side (pure Python integers) form of include/recogym_rng.h, for the policy callbacks that
run on the host in the per-user gym compatibility path (Agent.act).  The simulator itself never
draws on the host: every env draw happens in the HIP kernels."""
import math

M32 = 0xFFFFFFFF
DRAW_EVENT, DRAW_POLICY, DRAW_DRIFT, DRAW_RESET = 0, 1, 2, 3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def draw(seed, user, t, slot, purpose):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(int(user) & M32, int(t) & M32, int(slot) & M32, int(purpose) & M32,
                         seed & M32, seed >> 32)


def m53(a, b):
    return ((a >> 5) << 26) | (b >> 6)


def uniform(a, b):
    return m53(a, b) / 9007199254740992.0


def bounded(a, b, n):
    return (m53(a, b) * int(n)) >> 53


def policy_uniforms(seed, user, t):
    """The two uniforms and the bounded-int word pair of one policy act."""
    w = draw(seed, user, t, 0, DRAW_POLICY)
    return w, uniform(w[0], w[1]), uniform(w[2], w[3])
