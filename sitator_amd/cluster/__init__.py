"""Landmark-clustering plugins, located by name exactly like the reference does
(``importlib.import_module("..cluster." + name)``, ``LandmarkAnalysis.py:234``).

A plugin is a module exporting
``do_landmark_clustering(landmark_vectors, clustering_params, min_samples, verbose) -> dict`` with the
keys ``'cluster-size'``, ``'cluster-labels'``, ``'cluster-confs'`` and optionally
``'cluster-landmark-groupings'`` / ``'cluster-representative-lvecs'`` (``LandmarkAnalysis.py:89-93``).
``landmark_vectors`` is a ``LandmarkVectors`` handle on GPU-resident rows; ``np.asarray()`` of it
gives the dense matrix a third-party plugin may expect.
"""
