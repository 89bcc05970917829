"""Pin the oracle on every known-answer test the reference holds for the hot path
(SURVEY.md section 4): reference test/adaptertrimmer_test.cpp, test/filter_test.cpp,
test/polyx_test.cpp, test/sequence_test.cpp and editdistance_test()
(src/editdistance.cpp:141-172).  The vectors below are DATA from those tests."""
from fastplong_amd import abi, synth

READ102 = ("TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCCCGGGGAAATTTCCCGGGAAATTTCCCGGGATCGATCGATCGATCGAATTCC")


def test_trim_by_sequence_start(orc):
    # test/adaptertrimmer_test.cpp:4-11
    seq = "AGGTGCTGCGCATACTTTTCCACGGGGATACTACTGGGTGTTACCGTGGGAATGAATCCTTTTAACCTTAGCAATACGTAAAGGTGCT"
    adapter = "GCGCATACTTTTCCACGGGGATACTACTG"
    out, trimmed, key_len = orc.trim_start(seq, adapter, 0.3, 0)
    assert out == "GGTGTTACCGTGGGAATGAATCCTTTTAACCTTAGCAATACGTAAAGGTGCT"
    assert trimmed == 36 and key_len == len(adapter)


def test_trim_by_sequence_end(orc):
    # test/adaptertrimmer_test.cpp:13-18
    seq = "TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAAGCGCATACTTTTCCACGGGGA"
    adapter = "GCGCATACTTTTCCACGGGGATACTACTG"
    out, trimmed, key_len = orc.trim_end(seq, adapter, 0.3, 0)
    assert out == "TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAA"
    assert trimmed == 21


def test_search_adapter_left(orc):
    # test/adaptertrimmer_test.cpp:37-57
    assert orc.search_adapter(READ102, "TTTT", 0.3, 0, -1, True, False) == 0
    assert orc.search_adapter(READ102, "AACC", 0.3, 0, -1, True, False) == 4


def test_trim_and_cut(orc):
    # test/filter_test.cpp:4-22
    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=4, cut_front_quality=20,
                                 cut_tail_window=4, cut_tail_quality=20, trim_front=0, trim_tail=1)
    r = orc.trim_and_cut("TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTT", "/////CCCCCCCCCCCC////CCCCCCCCCCCCCC////E", opt)
    assert r is not None
    front, seq, qual = r
    assert seq == "CCCCCCCCCCCCCCCCCCCCCCCCCCCC"
    assert qual == "CCCCCCCCCCC////CCCCCCCCCCCCC"
    assert front == 6


def test_trim_polyx(orc):
    # test/polyx_test.cpp:4-17
    seq, called, poly, tl = orc.trim_polyx("ATTTTAAAAAAAAAATAAAAAAAAAAAAACAAAAAAAAAAAAAAAAAAAAAAAAAT",
                                           "///EEEEEEEEEEEEEEEEEEEEEEEEEE////EEEEEEEEEEEEE////E////E", 10)
    assert seq == "ATTTT"
    assert called == 1 and tl == 51 and poly == 0


def test_edit_distance_kat(orc):
    # src/editdistance.cpp:141-172 (editdistance_test): three 151-bp pairs -> 0, 1, 90
    s1 = [
        "CCTATCAGGGAGCTGTGGGCCAGCCAGGAGGCAGCACATGCCCAATCCCAGGCCCCTCCCGTTGTAAGTTCCCGTTCTACCCGACAGGGACCTGCTGACAAAAGACAGGGCTGGAGAGCCAGCCTGAAGGCCCTGGGACCCTTCTATCCAC",
        "ACTTATGTTTTTAAATGAGGATTATTGATAGTACTCTTGGTTTTTATACCATTCAGATCACTGAATTTATAAAGTACCCATCTAGTACTTCAAAAAGTAAAGTGTTCTGCCAGATCTTAGGTATAGAGGACCCTAACACAGTAAGATCGGA",
        "TAGGGGTATGAGTAGAGCTGAGCTGGGGGAAAAGAGGGAAATTCCCAGGGGTGGAGGAAGAGTCAAGTCCCCCTCTACACCTAGAGGATGAACTTAAGGAAGGAGTGAAGGTCATATGTGTTGTTCCTGAGGAAAAGGCCGCTGTAGAAAA",
    ]
    s2 = [
        "CCTATCAGGGAGCTGTGGGCCAGCCAGGAGGCAGCACATGCCCAATCCCAGGCCCCTCCCGTTGTAAGTTCCCGTTCTACCCGACAGGGACCTGCTGACAAAAGACAGGGCTGGAGAGCCAGCCTGAAGGCCCTGGGACCCTTCTATCCAC",
        "ACTTATGTTTTTAAATGAGGATTATTGATAGTACTCTTGGTTTTTATACCATTCAGATCACTGAATTTATAAAGTACCCATCTAGTACTTGAAAAAGTAAAGTGTTCTGCCAGATCTTAGGTATAGAGGACCCTAACACAGTAAGATCGGA",
        "CCTGGGCCTGGCCCTTGTCTAAAACTGACTCTTTTGAGGGTGATTTTGGATGTTCTTAGTAGAGTCTCTCACCTGTACTTTCCTTGCCTAAGGTGCTGTCTTCTCTTGCAGGTTGCCTACACGTTCCTCACATGCCCTAAGAACCATGGGA",
    ]
    for a, b, want in zip(s1, s2, [0, 1, 90]):
        assert orc.edit_distance(a, b) == want
        assert orc.edit_distance(b, a) == want


def test_reverse_complement():
    # test/sequence_test.cpp:4-9 (host helper used to derive the end adapter, src/main.cpp:138-140)
    assert synth.revcomp("AAAATTTTCCCCGGGG") == "CCCCGGGGAAAATTTT"
    assert synth.revcomp(synth.START_ADAPTER) == synth.END_ADAPTER
