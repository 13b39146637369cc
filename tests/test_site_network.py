"""CPU-only: the SiteNetwork data contract beyond what LandmarkAnalysis itself touches - subsets, single sites,
``update_centers`` (reference ``sitator/SiteNetwork.py:97-141,199-207,311-346``)."""
import pickle

import numpy as np
import pytest

from sitator_amd import SiteNetwork, Structure


def _network(n=6):
    rng = np.random.default_rng(0)
    st = Structure(rng.uniform(0, 10, (8, 3)), np.eye(3) * 10.0, numbers=[3, 3, 8, 8, 8, 8, 15, 15])
    sn = SiteNetwork(st, np.array([0, 0, 1, 1, 1, 1, 1, 1], bool), np.array([1, 1, 0, 0, 0, 0, 0, 0], bool))
    sn.centers = rng.uniform(0, 10, (n, 3))
    sn.vertices = [[i % 6, (i + 1) % 6, (i + 2) % 6] for i in range(n)]
    sn.site_types = np.arange(n) % 2
    sn.add_site_attribute("occupancy", np.linspace(0.1, 0.6, n))
    sn.add_edge_attribute("n_ij", np.arange(n * n, dtype=float).reshape(n, n))
    return sn


@pytest.mark.parametrize("key", [np.array([4, 1, 2]), np.array([True, False, True, True, False, False]), slice(1, 5, 2)])
def test_subset_cuts_every_per_site_and_per_edge_array(key):
    sn = _network()
    idx = np.arange(6)[key]
    sub = sn[key]
    assert sub.n_sites == len(idx) and sub.n_mobile == sn.n_mobile and sub.n_static == sn.n_static
    assert np.array_equal(sub.centers, sn.centers[idx])
    assert sub.vertices == [sn.vertices[i] for i in idx]
    assert np.array_equal(sub.site_types, sn.site_types[idx])
    assert np.array_equal(sub.occupancy, sn.occupancy[idx])
    assert np.array_equal(sub.n_ij, sn.n_ij[idx][:, idx])
    assert sorted(sub.site_attributes) == ["occupancy"] and sorted(sub.edge_attributes) == ["n_ij"]
    sub.update_centers(sub.centers + 1.0)                      # the subset owns its arrays
    assert not np.array_equal(sub.centers, sn.centers[idx])


def test_of_type_update_centers_get_site_get_edge():
    sn = _network()
    odd = sn.of_type(1)
    assert odd.n_sites == 3 and np.all(odd.site_types == 1) and np.array_equal(odd.centers, sn.centers[1::2])
    with pytest.raises(ValueError):
        sn.of_type(7)
    moved = sn.centers + 0.25
    sn.update_centers(moved)                                   # same sites: everything else stays
    assert np.array_equal(sn.centers, moved) and sn.vertices is not None and sn.has_attribute("occupancy")
    with pytest.raises(ValueError):
        sn.update_centers(moved[:3])
    site = sn.get_site(2)
    assert np.array_equal(site["center"], moved[2]) and site["vertices"] == [2, 3, 4] and site["type"] == 0
    assert site["occupancy"] == pytest.approx(0.3)
    assert sn.get_edge((1, 2)) == {"n_ij": 8.0}
    sn.centers = moved                                         # the setter is a new set of sites: the rest is dropped
    assert sn.vertices is None and not sn.has_attribute("occupancy")
    with pytest.raises(ValueError):
        sn.of_type(0)
    with pytest.raises(ValueError):
        sn.get_edge((0, 1))


def test_copy_and_pickle_keep_the_subset_consistent():
    sub = _network()[np.array([5, 0])]
    twin = pickle.loads(pickle.dumps(sub.copy()))
    assert np.array_equal(twin.centers, sub.centers) and np.array_equal(twin.n_ij, sub.n_ij) and twin.vertices == sub.vertices
