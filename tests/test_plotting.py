"""Plot helpers (matplotlib, Agg backend): they must draw without a display and put the right things on the axes."""
import numpy as np
import pytest

matplotlib = pytest.importorskip("matplotlib")
matplotlib.use("Agg")

import lightgbm_amd as lgb  # noqa: E402
from lightgbm_amd import rstyle  # noqa: E402


@pytest.fixture(scope="module")
def fitted():
    rng = np.random.RandomState(0)
    X = rng.randn(3000, 5)
    y = 2 * X[:, 0] + np.sin(X[:, 1]) + 0.1 * rng.randn(3000)
    d = lgb.Dataset(X, label=y, feature_name=["a", "b", "c", "d", "e"])
    ev = {}
    b = lgb.train({"device_type": "cpu", "verbose": -1, "num_leaves": 7, "metric": ["l2", "l1"]}, d, 15,
                  valid_sets=[d], evals_result=ev)
    return b, ev, X


def test_plot_importance_and_metric(fitted):
    b, ev, _ = fitted
    ax = lgb.plot_importance(b, max_num_features=2)
    assert len(ax.patches) == 2 and ax.get_yticklabels()[-1].get_text() == "a"
    ax = lgb.plot_importance(b, importance_type="gain", precision=1)
    assert ax.get_title() == "Feature importance"
    ax = lgb.plot_metric(ev, metric="l1")
    assert ax.get_ylabel() == "l1" and len(ax.lines) == 1 and len(ax.lines[0].get_ydata()) == 15
    with pytest.raises(KeyError):
        lgb.plot_metric(ev, metric="auc")


def test_split_value_histogram(fitted):
    b, _, _ = fitted
    hist, edges = b.get_split_value_histogram("a", bins=5)
    assert hist.sum() == sum(int(v) for v in b.feature_importance()[:1]) and len(edges) == 6
    xs = b.get_split_value_histogram(0, xgboost_style=True)
    assert xs.shape[1] == 2 and xs[:, 1].sum() == hist.sum()
    ax = lgb.plot_split_value_histogram(b, "a", bins=5)
    assert "name a" in ax.get_title()


def test_plot_tree_draws_every_node(fitted):
    b, _, _ = fitted
    ax = lgb.plot_tree(b, tree_index=2, show_info=["split_gain", "leaf_count"])
    n_leaves = b.trees_to_dataframe().query("tree_index == 2")["split_feature"].isna().sum()
    boxed = [t for t in ax.texts if t.get_bbox_patch() is not None]
    assert len(boxed) == 2 * n_leaves - 1
    with pytest.raises(IndexError):
        lgb.plot_tree(b, tree_index=99)
    try:
        import graphviz  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="graphviz"):
            lgb.create_tree_digraph(b)


def test_r_style_prepare_and_plots(fitted):
    import pandas as pd
    b, _, X = fitted
    df = pd.DataFrame({"cut": pd.Categorical(["Good", "Ideal", "Good", "Fair"], categories=["Fair", "Good", "Ideal"]),
                       "color": ["E", "D", "E", "J"], "carat": [0.2, 0.3, 0.25, 1.0]})
    res = rstyle.lgb_prepare_rules(df)
    assert res["data"]["cut"].tolist() == [2.0, 3.0, 2.0, 1.0] and res["data"]["color"].tolist() == [2.0, 1.0, 2.0, 3.0]
    assert res["rules"]["color"] == {"D": 1, "E": 2, "J": 3} and "carat" not in res["rules"]
    again = rstyle.lgb_prepare_rules(pd.DataFrame({"cut": ["Ideal", "Premium"], "color": ["J", "E"], "carat": [1.0, 2.0]}),
                                     rules=res["rules"])
    assert again["data"]["cut"].tolist()[0] == 3.0 and np.isnan(again["data"]["cut"].tolist()[1])
    assert rstyle.lgb_prepare(df).dtypes.map(lambda t: t.kind).tolist() == ["f", "f", "f"]
    imp = rstyle.lgb_importance(b)
    ax = rstyle.lgb_plot_importance(imp, top_n=2, measure="Gain")
    assert len(ax.patches) == 2
    inter = rstyle.lgb_interprete(b, X, [1])[0]
    ax = rstyle.lgb_plot_interpretation(inter, top_n=3)
    assert len(ax.patches) == 3
    d = rstyle.lgb_Dataset(X, label=X[:, 0])
    rstyle.lgb_Dataset_set_categorical(d, [4])
    assert d.categorical_feature == [4] and rstyle.dimnames(d)[1][0] == "Column_0"
