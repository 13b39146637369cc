"""Plugin ``"dotprod"``: the online leader clustering of the landmark-analysis paper
(reference ``sitator/landmark/cluster/dotprod.py:6-33``), on the GPU."""
from ..dotprod_classifier import DotProdClassifier

DEFAULT_PARAMS = {
    "clustering_threshold": 0.45,
    "assignment_threshold": 0.8,
}


def do_landmark_clustering(landmark_vectors, clustering_params, min_samples, verbose):
    params = dict(DEFAULT_PARAMS)
    params.update(clustering_params)
    clf = DotProdClassifier(threshold=params["clustering_threshold"], min_samples=min_samples)
    labels, confs = clf.fit_predict(landmark_vectors, predict_threshold=params["assignment_threshold"],
                                    verbose=verbose)
    if hasattr(landmark_vectors, "ctx"):     # the centres the labels were assigned with (tests compare against the oracle)
        landmark_vectors.assignment = {"centers": clf.cluster_centers, "normed": True, "threshold": params["assignment_threshold"]}
    return {
        "cluster-size": clf.cluster_counts,
        "cluster-labels": labels,
        "cluster-confs": confs,
        "cluster-representative-lvecs": clf.cluster_centers,
    }
