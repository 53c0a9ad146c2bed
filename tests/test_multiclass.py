"""Multiclass objectives (softmax and one-vs-all): num_class trees per iteration.

Upstream LightGBM is not installed here, so its numbers cannot be pinned; quality is checked against
scikit-learn's histogram gradient boosting on the same data, the metrics against their definitions."""
import numpy as np
import pytest

import lightgbm_amd as lgb
from lightgbm_amd.models import TreeEnsemble
from lightgbm_amd.parallel import ThreadComm

CPU = {"device_type": "cpu", "verbose": -1}


def make_classes(n, K=3, F=6, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, F)
    W = rng.randn(F, K)
    s = X @ W + 0.5 * np.sin(2 * X[:, :K]) + 0.4 * rng.randn(n, K)
    return X, s.argmax(axis=1).astype(np.float64)


def test_softmax_training_metrics_and_prediction_shapes():
    X, y = make_classes(6000, K=4)
    d = lgb.Dataset(X, label=y, params=CPU)
    ev = {}
    p = {**CPU, "objective": "multiclass", "num_class": 4, "num_leaves": 15, "metric": ["multi_logloss", "multi_error"]}
    b = lgb.train(p, d, 30, valid_sets=[d], evals_result=ev)
    assert b.num_trees() == 120 and b.current_iteration() == 30 and b.num_model_per_iteration() == 4
    prob = b.predict(X)
    raw = b.predict(X, raw_score=True)
    assert prob.shape == raw.shape == (6000, 4)
    np.testing.assert_allclose(prob.sum(axis=1), 1.0, atol=1e-12)
    e = np.exp(raw - raw.max(axis=1, keepdims=True))
    np.testing.assert_allclose(prob, e / e.sum(axis=1, keepdims=True), atol=1e-12)
    yi = y.astype(int)
    # the recorded metrics are the definitions applied to the predictions
    assert abs(ev["training"]["multi_logloss"][-1] + np.mean(np.log(prob[np.arange(6000), yi]))) < 1e-9
    assert abs(ev["training"]["multi_error"][-1] - np.mean(prob.argmax(axis=1) != yi)) < 1e-9
    assert ev["training"]["multi_logloss"][-1] < 0.5 * ev["training"]["multi_logloss"][0]
    # staged prediction, leaf indices, contributions
    np.testing.assert_array_equal(b.predict(X[:50], num_iteration=10, raw_score=True),
                                  lgb.Booster(model_str=b.model_to_string(num_iteration=10)).predict(X[:50], raw_score=True))
    assert b.predict(X[:7], pred_leaf=True).shape == (7, 120)
    phi = b.predict(X[:7], pred_contrib=True).reshape(7, 4, 7)
    np.testing.assert_allclose(phi.sum(axis=2), raw[:7], atol=1e-10)
    # text model: header, round trip, independent NumPy reader
    text = b.model_to_string()
    assert "num_class=4\n" in text and "num_tree_per_iteration=4\n" in text and "objective=multiclass num_class:4\n" in text
    again = lgb.Booster(model_str=text)
    np.testing.assert_array_equal(again.predict(X), prob)
    assert again.model_to_string() == text
    np.testing.assert_allclose(TreeEnsemble.from_string(text).predict(X), prob, atol=1e-12)
    assert b.dump_model()["num_tree_per_iteration"] == 4 and len(b.dump_model()["tree_info"]) == 120


def test_first_iteration_starts_from_the_class_priors():
    X, y = make_classes(3000, K=3, seed=2)
    y[:1200] = 0.0     # unbalanced
    b = lgb.train({**CPU, "objective": "multiclass", "num_class": 3, "num_leaves": 2, "learning_rate": 1e-9,
                   "min_gain_to_split": 1e9}, lgb.Dataset(X, label=y, params=CPU), 1)
    prior = np.bincount(y.astype(int), minlength=3) / len(y)
    np.testing.assert_allclose(b.predict(X[:3]), np.tile(prior, (3, 1)), atol=1e-9)


def test_quality_is_on_par_with_sklearn_hist_gradient_boosting():
    from sklearn.ensemble import HistGradientBoostingClassifier
    X, y = make_classes(12000, K=5, F=8, seed=3)
    Xt, yt, Xv, yv = X[:9000], y[:9000], X[9000:], y[9000:]
    b = lgb.train({**CPU, "objective": "multiclass", "num_class": 5, "num_leaves": 31, "learning_rate": 0.1},
                  lgb.Dataset(Xt, label=yt, params=CPU), 60)
    sk = HistGradientBoostingClassifier(max_iter=60, learning_rate=0.1, max_leaf_nodes=31, min_samples_leaf=20,
                                        early_stopping=False).fit(Xt, yt)
    ours = np.mean(b.predict(Xv).argmax(axis=1) == yv)
    theirs = np.mean(sk.predict(Xv) == yv)
    assert ours > theirs - 0.015, (ours, theirs)


def test_one_vs_all_rollback_custom_gradients_and_early_stopping():
    X, y = make_classes(4000, K=3, seed=4)
    d = lgb.Dataset(X, label=y, params=CPU)
    p = {**CPU, "objective": "multiclassova", "num_class": 3, "num_leaves": 15}
    b = lgb.train(p, d, 25, keep_training_booster=True)
    prob = b.predict(X)
    assert prob.shape == (4000, 3) and np.mean(prob.argmax(axis=1) == y) > 0.85
    raw = b.predict(X, raw_score=True)
    np.testing.assert_allclose(prob, 1 / (1 + np.exp(-raw)), atol=1e-12)
    assert "objective=multiclassova num_class:3 sigmoid:1\n" in b.model_to_string()
    # training scores == model predictions; rolling back removes one tree per class from both
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), raw, atol=1e-9)
    b.rollback_one_iter()
    assert b.num_trees() == 72
    np.testing.assert_allclose(b._inner_predict(-1, raw=True), b.predict(X, raw_score=True), atol=1e-9)

    # softmax through caller-supplied gradients == the built-in objective
    def softmax_obj(preds, data):
        lab = data.get_label().astype(int)
        e = np.exp(preds - preds.max(axis=1, keepdims=True))
        pr = e / e.sum(axis=1, keepdims=True)
        g = pr.copy()
        g[np.arange(len(lab)), lab] -= 1.0
        return g, 1.5 * pr * (1.0 - pr)
    q = {**CPU, "num_class": 3, "num_leaves": 7, "boost_from_average": False}
    c = lgb.train({**q, "objective": "multiclass"}, lgb.Dataset(X, label=y, params=CPU), 5)
    e = lgb.train(q, lgb.Dataset(X, label=y, params=CPU), 5, fobj=softmax_obj)
    np.testing.assert_allclose(c.predict(X, raw_score=True), e.predict(X, raw_score=True), atol=1e-5)

    # early stopping + cv on the multiclass metric
    dv = d.create_valid(X[:800], label=np.roll(y[:800], 1))     # labels unrelated to the features
    s = lgb.train({**CPU, "objective": "multiclass", "num_class": 3}, d, 200, valid_sets=[dv], early_stopping_rounds=3,
                  verbose_eval=False)
    assert 0 < s.best_iteration < 60 and s.predict(X[:5]).shape == (5, 3)
    r = lgb.cv({**CPU, "objective": "multiclass", "num_class": 3, "num_leaves": 7}, d, 15, nfold=3, stratified=True)
    assert r["multi_logloss-mean"][-1] < r["multi_logloss-mean"][0]


def test_refused_configurations():
    X, y = make_classes(500, K=3)
    with pytest.raises(Exception, match="num_class"):
        lgb.train({**CPU, "objective": "multiclass"}, lgb.Dataset(X, label=y, params=CPU), 2)
    with pytest.raises(Exception, match="num_class"):
        lgb.train({**CPU, "objective": "regression", "num_class": 3}, lgb.Dataset(X, label=y, params=CPU), 2)
    with pytest.raises(Exception, match="labels"):
        lgb.train({**CPU, "objective": "multiclass", "num_class": 2}, lgb.Dataset(X, label=y, params=CPU), 2)


def test_sklearn_classifier_with_string_labels():
    from lightgbm_amd.sklearn import LGBMClassifier
    X, y = make_classes(3000, K=3, seed=6)
    names = np.array(["ant", "bee", "cat"])[y.astype(int)]
    c = LGBMClassifier(n_estimators=25, num_leaves=15, device_type="cpu").fit(X, names)
    assert list(c.classes_) == ["ant", "bee", "cat"]
    assert c.predict_proba(X).shape == (3000, 3) and np.mean(c.predict(X) == names) > 0.85


def test_data_parallel_multiclass_equals_serial():
    import threading
    X, y = make_classes(4000, K=3, seed=7)
    p = {**CPU, "objective": "multiclass", "num_class": 3, "num_leaves": 15}
    serial = lgb.train(p, lgb.Dataset(X, label=y, bin_sample=X), 1)
    comms = ThreadComm.make_group(2)
    out, err = [None, None], [None, None]

    def work(r):
        try:
            sl = slice(0, 2000) if r == 0 else slice(2000, None)
            out[r] = lgb.train({**p, "tree_learner": "data"}, lgb.Dataset(X[sl], label=y[sl], bin_sample=X), 1,
                               comm=comms[r]).model_to_string()
        except BaseException as e:   # noqa: a failing rank must not leave the other one waiting at the barrier
            err[r] = e
            comms[r]._hub.barrier.abort()
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    [c.close() for c in comms]
    cut = lambda s: s[s.index("Tree=0"):s.index("end of trees")]
    assert out[0] is not None and cut(out[0]) == cut(out[1]) == cut(serial.model_to_string())


def test_auc_mu_is_the_mean_pairwise_auc():
    from sklearn.metrics import roc_auc_score
    X, y = make_classes(4000, K=4, seed=6)
    d = lgb.Dataset(X, label=y, params=CPU)
    ev = {}
    b = lgb.train({**CPU, "objective": "multiclass", "num_class": 4, "num_leaves": 7, "metric": ["multi_logloss", "auc_mu"]}, d, 8,
                  valid_sets=[d], evals_result=ev)
    raw = b.predict(X, raw_score=True)
    pairs = []
    for i in range(4):
        for j in range(i + 1, 4):
            m = (y == i) | (y == j)
            pairs.append(roc_auc_score(y[m] == j, raw[m, j] - raw[m, i]))
    assert list(ev["training"]) == ["multi_logloss", "auc_mu"]
    assert ev["training"]["auc_mu"][-1] == pytest.approx(np.mean(pairs), abs=1e-9)


def test_auc_mu_weights_and_multi_error_top_k():
    from sklearn.metrics import roc_auc_score
    K = 4
    X, y = make_classes(4000, K=K, seed=7)
    A = np.array([[0, 1, 2, 3], [1, 0, 1, 2], [2, 1, 0, 1], [3, 2, 1, 0]], dtype=float)   # ordinal classes: far misses cost more
    d = lgb.Dataset(X, label=y, params=CPU)
    ev = {}
    p = {**CPU, "objective": "multiclass", "num_class": K, "num_leaves": 7, "metric": ["auc_mu", "multi_error"],
         "auc_mu_weights": A.ravel().tolist(), "multi_error_top_k": 2}
    b = lgb.train(p, d, 8, valid_sets=[d], evals_result=ev)
    raw = b.predict(X, raw_score=True)
    pairs = []
    for i in range(K):
        for j in range(i + 1, K):
            m = (y == i) | (y == j)
            pairs.append(roc_auc_score(y[m] == j, raw[m] @ (A[i] - A[j])))
    assert ev["training"]["auc_mu"][-1] == pytest.approx(np.mean(pairs), abs=1e-9)
    # top-2 error: wrong only when two other classes score at least as high as the true one
    true_score = raw[np.arange(len(y)), y.astype(int)]
    wrong = ((raw >= true_score[:, None]).sum(axis=1) > 2).mean()
    assert ev["training"]["multi_error"][-1] == pytest.approx(wrong, abs=1e-12)
    top1 = lgb.train({**p, "multi_error_top_k": 1}, d, 8, valid_sets=[d], evals_result=ev)
    assert ev["training"]["multi_error"][-1] == pytest.approx((raw.argmax(axis=1) != y).mean(), abs=1e-12) and top1 is not None
    with pytest.raises(lgb.LightGBMError):
        lgb.train({**p, "auc_mu_weights": [0.0, 1.0, 1.0]}, d, 1, valid_sets=[d])


def test_continued_training_of_a_multiclass_model():
    # 4 + 4 rounds through init_model equal 8 rounds in one go (no bagging, nothing random in between)
    rng = np.random.RandomState(11)
    X = rng.randn(900, 6)
    y = (np.digitize(X[:, 0] + 0.5 * X[:, 1], [-0.6, 0.6])).astype(float)
    p = {"device_type": "cpu", "verbose": -1, "objective": "multiclass", "num_class": 3, "min_data_in_leaf": 10,
         "num_leaves": 7}
    whole = lgb.train(p, lgb.Dataset(X, label=y), 8)
    first = lgb.train(p, lgb.Dataset(X, label=y), 4)
    d = lgb.Dataset(X, label=y, free_raw_data=False)
    dv = lgb.Dataset(X[:300], label=y[:300], reference=d, free_raw_data=False)
    ev = {}
    more = lgb.train(p, d, 4, init_model=first, valid_sets=[dv], evals_result=ev)
    assert more.num_trees() == 24
    np.testing.assert_allclose(more.predict(X), whole.predict(X), rtol=1e-6, atol=1e-9)
    pr = more.predict(X[:300])
    ll = -np.mean(np.log(pr[np.arange(300), y[:300].astype(int)]))
    assert ev["valid_0"]["multi_logloss"][-1] == pytest.approx(ll, rel=1e-9)
    # per-class initial scores given directly: [rows, classes]
    init = first.predict(X, raw_score=True)
    d2 = lgb.Dataset(X, label=y, init_score=init)
    b2 = lgb.train(p, d2, 4)
    np.testing.assert_allclose(b2.predict(X, raw_score=True) + init, more.predict(X, raw_score=True), rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(d2.get_init_score().reshape(3, -1).T, init)
