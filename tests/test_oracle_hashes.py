"""Pin Keccak/SHAKE256, Merlin and ChaCha in the oracle against independent vectors."""
import ctypes
import hashlib


def test_shake256_vs_hashlib(oracle):
    for msg in (b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 137, bytes(range(256)) * 3):
        for m in (1, 32, 136, 137, 500):
            out = (ctypes.c_uint8 * m)()
            oracle.orc_shake256(msg, ctypes.c_size_t(len(msg)), out, ctypes.c_size_t(m))
            assert bytes(out) == hashlib.shake_256(msg).digest(m)


def test_merlin_published_vector(oracle):
    # merlin 3.0.0 src/transcript.rs tests (`equivalence_simple` inputs) / merlin.cool conformance vector:
    # Transcript::new(b"test protocol"); append_message(b"some label", b"some data"); challenge_bytes(b"challenge", 32)
    out = (ctypes.c_uint8 * 32)()
    oracle.orc_merlin_simple(b"test protocol", b"some label", b"some data", ctypes.c_size_t(9), b"challenge", out, ctypes.c_size_t(32))
    assert bytes(out).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_chacha20_rfc7539_block(oracle):
    # RFC 7539 §2.3.2 uses a 32-bit counter + 96-bit nonce; with nonce = 0 and counter = 1 the state equals
    # rand_chacha's (64-bit counter, 64-bit stream id = 0) layout, so the keystream block must agree with
    # the IETF reference computed here in pure Python.
    key = bytes(range(32))

    def rotl(x, n):
        return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF

    def block(key, counter, rounds):
        st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)] + [counter & 0xFFFFFFFF, counter >> 32, 0, 0]
        x = list(st)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

        for _ in range(rounds // 2):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return [(x[i] + st[i]) & 0xFFFFFFFF for i in range(16)]

    for rounds in (20, 12):
        for ctr in (0, 1, 2**32 + 5):
            out = (ctypes.c_uint32 * 16)()
            oracle.orc_chacha_block(key, ctypes.c_uint64(ctr), rounds, out)
            assert list(out) == block(key, ctr, rounds)
    # RFC 7539 §2.3.2 first output word for key 00..1f, counter 1, nonce (0,0x4a000000,0) differs in nonce, so only
    # the zero-nonce self-consistency above is asserted; the quarter-round KAT of §2.1.1 pins the round function:
    a, b, c, d = 0x11111111, 0x01020304, 0x9B8D6F43, 0x01234567
    a = (a + b) & 0xFFFFFFFF; d = rotl(d ^ a, 16); c = (c + d) & 0xFFFFFFFF; b = rotl(b ^ c, 12)
    a = (a + b) & 0xFFFFFFFF; d = rotl(d ^ a, 8); c = (c + d) & 0xFFFFFFFF; b = rotl(b ^ c, 7)
    assert (a, b, c, d) == (0xEA2A92F4, 0xCB1CF8CE, 0x4581472E, 0x5881C4BB)
