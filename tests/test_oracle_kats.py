"""Every known answer the reference's own unit tests hold for the hot path, run against the oracle,
plus the reference's acceptance criterion (prove -> verify) on its four e2e configurations.  The two `spark_unconfirmed` entries are NOT reference tests: the strategy is not in the
snapshot; they hold the restatement to the self-consistency properties the reference's macro checks for every strategy (MLE = table on the hypercube) and to prove -> verify."""
import pytest

NAMES = ("poly_evaluation_28,poly_evaluation_const8,eq_evals_vs_naive,unipoly_quad,unipoly_cubic,gauss,split_bits,grand_product_24,"
         "sumcheck_scripted_313,memory_checking_multiset,and_table,and_merged_poly,or_table,xor_table,lt_table,range_table,"
         "poly_commit_open_verify,dot_product_log,e2e_prove_4d_lt,e2e_prove_4d_lt_big_s,e2e_prove_4d_and,e2e_prove_3d_range,"
         "e2e_prove_1d_and_s64,e2e_prove_2d_xor,e2e_prove_2d_or,spark_unconfirmed,e2e_prove_spark_unconfirmed,rand_stdrng_value_stability,rand_chacha20_true_values_a").split(",")


def test_names_in_sync(oracle):
    assert oracle.orc_kat_names().decode().split(",") == NAMES


@pytest.mark.parametrize("name", NAMES)
def test_reference_kat(oracle, name):
    assert getattr(oracle, "orc_kat_" + name)() == 0
