"""Host-side rows of SURVEY.md 8(f)-4 against vectors produced by the reference itself (tests/golden/make_golden.py):
  * oracle/kmeans_oracle.py == code/models/kmeans/kmeans.py under random.seed: starting centroids, every iteration's
    assignments, final centroids / masks / distances / clustering_results (kmeans.npz) -- five cases incl. clusters that
    run empty ('singleton' refill and the reference's centroid-is-a-view-of-X behaviour), an unobserved column, the toy
    BNMTF matrix by rows and by columns (what initialise(init_FG='kmeans') clusters);
  * bnmtf_amd.data.load_gdsc / negate_gdsc == data_drug_sensitivity/gdsc/load_data.py:15-70 on the first 12 cell lines of the
    reference's own ic50 file (the excerpt is a fixture), and on the whole file where /root/reference is present."""
import os

import numpy as np
import pytest

from bnmtf_amd import data
from oracle.kmeans_oracle import KMeansOracle

HERE = os.path.dirname(os.path.abspath(__file__))
KM = np.load(os.path.join(HERE, "golden", "kmeans.npz"))
CASES = sorted(set(k.split("/")[0] for k in KM.files))


@pytest.mark.parametrize("name", CASES)
def test_kmeans_oracle_reproduces_the_reference(name):
    g = {k.split("/", 1)[1]: KM[k] for k in KM.files if k.startswith(name + "/")}
    km = KMeansOracle(g["X"], g["M"], int(g["K"]))
    km.initialise(int(g["seed"]))
    assert np.array_equal(np.array(km.centroids), g["centroids0"])                 # same random.uniform call order
    km.cluster()
    assert np.array_equal(np.array(km.assign_hist), g["assign_hist"])              # every iteration, not only the last
    np.testing.assert_allclose(np.array(km.centroids), g["centroids"], rtol=1e-13, atol=1e-13)
    assert np.array_equal(km.mask_centroids, g["mask_centroids"])
    np.testing.assert_allclose(km.distances, g["distances"], rtol=1e-12, atol=1e-12, equal_nan=True)
    assert np.array_equal(km.clustering_results, g["clustering_results"])


def test_kmeans_golden_cases_cover_the_empty_cluster_rule():
    """At least one case refills an empty cluster (an assignment history in which a cluster's only member is the point the
    'singleton' rule moved): otherwise the cases above would not pin kmeans.py:137-152."""
    g = KM["empties/assign_hist"]
    assert len(set(g[-1])) == int(KM["empties/K"])          # all six clusters are populated although the data has three groups


def test_gdsc_loader_matches_the_reference_on_its_own_file():
    g = np.load(os.path.join(HERE, "golden", "gdsc.npz"))
    X, X_min, M, drugs, cells, cancers, tissues = data.load_gdsc(os.path.join(HERE, "golden", "gdsc_excerpt.txt"))
    assert np.array_equal(X, g["ex/X"]) and np.array_equal(X_min, g["ex/X_min"]) and np.array_equal(M, g["ex/M"])
    assert len(drugs) == int(g["ex/n_drugs"]) and len(cells) == int(g["ex/n_cells"]) == len(cancers) == len(tissues)
    assert np.array_equal(data.negate_gdsc(X, M), g["ex/negated"])
    full = "/root/reference/data_drug_sensitivity/gdsc/ic50_excl_empty_filtered_cell_lines_drugs.txt"
    if not os.path.exists(full):
        pytest.skip("the whole file is only in the build container")
    X, X_min, M, drugs, cells, cancers, tissues = data.load_gdsc(full)
    assert tuple(X.shape) == tuple(g["full/shape"]) and M.sum() == float(g["full/M_sum"])
    assert X.sum() == float(g["full/X_sum"]) and X.min() == float(g["full/minimum"])
    np.testing.assert_allclose(X_min.sum(), float(g["full/X_min_sum"]), rtol=1e-13)
    assert np.array_equal(M.sum(axis=1), g["full/row_obs"]) and np.array_equal(M.sum(axis=0), g["full/col_obs"])
    ii, jj = g["full/ii"], g["full/jj"]
    assert np.array_equal(X[ii, jj], g["full/X_at"]) and np.array_equal(M[ii, jj], g["full/M_at"]) and np.array_equal(X_min[ii, jj], g["full/X_min_at"])
