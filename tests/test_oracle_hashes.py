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


def test_chacha_published_vectors_8_12_20_rounds(oracle):
    """draft-strombergson-chacha-test-vectors-01 (256-bit keys, IV = 0, first keystream block): TC1 (all-zero key) for 8, 12 and 20 rounds and
    TC2 (key = 01 00 .. 00) for 12 rounds.  ark_std::test_rng() is rand_chacha's StdRng = ChaCha12 (ark-std 0.4 `test_rng`, rand 0.8), so the
    12-round rows pin the oracle's `ChaChaRng(seed, 12)` block function to a PUBLISHED answer (round 1 had only the 20-round one)."""
    tc1 = {8: "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42",
           12: "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be",
           20: "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586"}
    for rounds, want in tc1.items():
        out = (ctypes.c_uint32 * 16)()
        oracle.orc_chacha_block(bytes(32), ctypes.c_uint64(0), rounds, out)
        assert bytes(out).hex() == want
    out = (ctypes.c_uint32 * 16)()
    oracle.orc_chacha_block(bytes([1] + [0] * 31), ctypes.c_uint64(0), 12, out)
    assert bytes(out).hex() == ("12056e595d56b0f6eef090f0cd25a20949248c2790525d0f930218ff0b4ddd10"
                                "a6002239d9a454e29e107a7d06fefdfef0210feba044f9f29b1772c960dc29c0")
