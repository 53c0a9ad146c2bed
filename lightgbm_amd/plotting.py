"""Plots of a trained model (matplotlib): feature importance, recorded metrics, split-value histograms and
single trees.  ``create_tree_digraph`` needs the ``graphviz`` package; ``plot_tree`` draws with matplotlib
alone."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np

from .basic import Booster, LightGBMError

__all__ = ["plot_importance", "plot_metric", "plot_split_value_histogram", "plot_tree", "create_tree_digraph"]


def _axes(ax, figsize, dpi):
    import matplotlib.pyplot as plt
    if ax is None:
        _, ax = plt.subplots(1, 1, figsize=figsize, dpi=dpi)
    return ax


def _booster_of(model) -> Booster:
    if isinstance(model, Booster):
        return model
    if hasattr(model, "booster_"):
        return model.booster_
    raise TypeError("booster must be a Booster or a fitted LGBMModel")


def plot_importance(booster, ax=None, height: float = 0.2, xlim=None, ylim=None, title: Optional[str] = "Feature importance",
                    xlabel: Optional[str] = "Feature importance", ylabel: Optional[str] = "Features",
                    importance_type: str = "split", max_num_features: Optional[int] = None, ignore_zero: bool = True,
                    figsize=None, dpi=None, grid: bool = True, precision: Optional[int] = 3, **kwargs):
    """Horizontal bars of the features' split counts (``importance_type="split"``) or total gains (``"gain"``)."""
    b = _booster_of(booster)
    imp = np.asarray(b.feature_importance(importance_type=importance_type), dtype=np.float64)
    names = b.feature_name()
    pairs = sorted(zip(names, imp), key=lambda p: p[1])
    if ignore_zero:
        pairs = [p for p in pairs if p[1] != 0]
    if max_num_features is not None and max_num_features > 0:
        pairs = pairs[-max_num_features:]
    if not pairs:
        raise ValueError("Booster's feature_importance is empty.")
    labels, values = zip(*pairs)
    ax = _axes(ax, figsize, dpi)
    ylocs = np.arange(len(values))
    ax.barh(ylocs, values, align="center", height=height, **kwargs)
    for x, y in zip(values, ylocs):
        ax.text(x + 1, y, f"{x:.{precision}f}" if importance_type == "gain" and precision is not None else f"{int(x)}" if importance_type == "split" else str(x),
                va="center")
    ax.set_yticks(ylocs)
    ax.set_yticklabels(labels)
    ax.set_xlim(xlim if xlim is not None else (0, max(values) * 1.1))
    ax.set_ylim(ylim if ylim is not None else (-1, len(values)))
    if title is not None:
        ax.set_title(title)
    if xlabel is not None:
        ax.set_xlabel(xlabel)
    if ylabel is not None:
        ax.set_ylabel(ylabel)
    ax.grid(grid)
    return ax


def plot_metric(booster: Union[Dict[str, Dict[str, List[float]]], Any], metric: Optional[str] = None,
                dataset_names: Optional[List[str]] = None, ax=None, xlim=None, ylim=None,
                title: Optional[str] = "Metric during training", xlabel: Optional[str] = "Iterations",
                ylabel: Optional[str] = "@metric@", figsize=None, dpi=None, grid: bool = True):
    """One curve per dataset of a metric recorded with ``record_evaluation`` / ``evals_result`` (or a fitted
    estimator's ``evals_result_``)."""
    if isinstance(booster, dict):
        results = booster
    elif hasattr(booster, "evals_result_"):
        results = booster.evals_result_
    else:
        raise TypeError("booster must be a dict of recorded evaluations or a fitted LGBMModel")
    if not results:
        raise ValueError("eval results cannot be empty.")
    names = list(dataset_names) if dataset_names else list(results)
    first = results[names[0]]
    if metric is None:
        metric = next(iter(first))
    if metric not in first:
        raise KeyError(f"No given metric {metric!r} in eval results.")
    ax = _axes(ax, figsize, dpi)
    n_it = 0
    for name in names:
        vals = results[name][metric]
        n_it = max(n_it, len(vals))
        ax.plot(range(len(vals)), vals, label=name)
    ax.legend(loc="best")
    ax.set_xlim(xlim if xlim is not None else (0, n_it))
    if ylim is not None:
        ax.set_ylim(ylim)
    if title is not None:
        ax.set_title(title)
    if xlabel is not None:
        ax.set_xlabel(xlabel)
    if ylabel is not None:
        ax.set_ylabel(ylabel.replace("@metric@", metric))
    ax.grid(grid)
    return ax


def plot_split_value_histogram(booster, feature: Union[int, str], bins=None, ax=None, width_coef: float = 0.8, xlim=None,
                               ylim=None, title: Optional[str] = "Split value histogram for feature with @index/name@ @feature@",
                               xlabel: Optional[str] = "Feature split value", ylabel: Optional[str] = "Count",
                               figsize=None, dpi=None, grid: bool = True, **kwargs):
    """Histogram of the thresholds the model uses on one feature."""
    b = _booster_of(booster)
    hist, edges = b.get_split_value_histogram(feature, bins=bins)
    if np.count_nonzero(hist) == 0:
        raise ValueError(f"Cannot plot split value histogram, because feature {feature} was not used in splitting")
    ax = _axes(ax, figsize, dpi)
    width = width_coef * (edges[1] - edges[0])
    centres = (edges[1:] + edges[:-1]) / 2.0
    ax.bar(centres, hist, align="center", width=width, **kwargs)
    if xlim is None:
        span = (edges[-1] - edges[0]) * 0.2
        xlim = (edges[0] - span, edges[-1] + span)
    ax.set_xlim(xlim)
    ax.set_ylim(ylim if ylim is not None else (0, max(hist) * 1.1))
    if title is not None:
        ax.set_title(title.replace("@feature@", str(feature)).replace("@index/name@", "name" if isinstance(feature, str) else "index"))
    if xlabel is not None:
        ax.set_xlabel(xlabel)
    if ylabel is not None:
        ax.set_ylabel(ylabel)
    ax.grid(grid)
    return ax


def _tree_structure(b: Booster, tree_index: int) -> Tuple[Dict[str, Any], List[str]]:
    model = b.dump_model()
    if not 0 <= tree_index < len(model["tree_info"]):
        raise IndexError("tree_index is out of range.")
    return model["tree_info"][tree_index]["tree_structure"], model.get("feature_names") or []


def _node_text(node: Dict[str, Any], names: List[str], show_info: List[str], precision: Optional[int]) -> str:
    fmt = (lambda v: f"{v:.{precision}f}") if precision is not None else str
    if "split_feature" in node:
        f = node["split_feature"]
        name = names[f] if f < len(names) else f"Column_{f}"
        op = node.get("decision_type", "<=")
        thr = node["threshold"]
        text = f"{name} {op} {fmt(thr) if isinstance(thr, float) else thr}"
        for key in ("split_gain", "internal_value", "internal_count", "internal_weight"):
            if key in show_info and key in node:
                text += f"\n{key}: {fmt(node[key]) if isinstance(node[key], float) else node[key]}"
        return text
    text = f"leaf {node.get('leaf_index', 0)}: {fmt(node['leaf_value'])}"
    for key in ("leaf_count", "leaf_weight"):
        if key in show_info and key in node:
            text += f"\n{key}: {fmt(node[key]) if isinstance(node[key], float) else node[key]}"
    return text


def plot_tree(booster, ax=None, tree_index: int = 0, figsize=None, dpi=None, show_info: Optional[List[str]] = None,
              precision: Optional[int] = 3, orientation: str = "horizontal"):
    """Draws one tree: leaves are spread evenly, every split sits between its children."""
    b = _booster_of(booster)
    root, names = _tree_structure(b, tree_index)
    show_info = list(show_info or [])
    ax = _axes(ax, figsize, dpi)
    positions: List[Tuple[float, float, str, bool]] = []
    edges: List[Tuple[Tuple[float, float], Tuple[float, float], str]] = []
    next_leaf = [0]

    def place(node, depth) -> Tuple[float, float]:
        if "split_feature" not in node:
            pos = (float(depth), float(next_leaf[0]))
            next_leaf[0] += 1
            positions.append((*pos, _node_text(node, names, show_info, precision), True))
            return pos
        left = place(node["left_child"], depth + 1)
        right = place(node["right_child"], depth + 1)
        pos = (float(depth), (left[1] + right[1]) / 2.0)
        positions.append((*pos, _node_text(node, names, show_info, precision), False))
        edges.append((pos, left, "yes"))
        edges.append((pos, right, "no"))
        return pos

    place(root, 0)
    swap = orientation == "vertical"
    xy = (lambda p: (p[1], -p[0])) if swap else (lambda p: (p[0], -p[1]))
    for a, c, label in edges:
        (x0, y0), (x1, y1) = xy(a), xy(c)
        ax.plot([x0, x1], [y0, y1], color="0.5", linewidth=1, zorder=1)
        ax.text((x0 + x1) / 2, (y0 + y1) / 2, label, fontsize=7, color="0.3", ha="center", va="center")
    for d, pos, text, is_leaf in positions:
        x, y = xy((d, pos))
        ax.text(x, y, text, ha="center", va="center", fontsize=8, zorder=2,
                bbox=dict(boxstyle="round" if is_leaf else "square", facecolor="#e8f4e8" if is_leaf else "#e8eef8", edgecolor="0.4"))
    xs = [xy((d, p))[0] for d, p, _, _ in positions]
    ys = [xy((d, p))[1] for d, p, _, _ in positions]
    ax.set_xlim(min(xs) - 0.6, max(xs) + 0.6)
    ax.set_ylim(min(ys) - 0.6, max(ys) + 0.6)
    ax.axis("off")
    return ax


def create_tree_digraph(booster, tree_index: int = 0, show_info: Optional[List[str]] = None, precision: Optional[int] = 3,
                        orientation: str = "horizontal", **kwargs):
    """The tree as a ``graphviz.Digraph`` (needs the graphviz package)."""
    try:
        from graphviz import Digraph
    except ImportError as e:
        raise ImportError("You must install graphviz to create a tree digraph; plot_tree draws with matplotlib alone.") from e
    b = _booster_of(booster)
    root, names = _tree_structure(b, tree_index)
    show_info = list(show_info or [])
    graph = Digraph(**kwargs)
    graph.attr("graph", nodesep="0.05", ranksep="0.3", rankdir="LR" if orientation == "horizontal" else "TB")
    counter = [0]

    def add(node, parent=None, decision=None):
        name = f"n{counter[0]}"
        counter[0] += 1
        graph.node(name, label=_node_text(node, names, show_info, precision),
                   shape="ellipse" if "split_feature" not in node else "rectangle")
        if parent is not None:
            graph.edge(parent, name, decision)
        if "split_feature" in node:
            add(node["left_child"], name, "yes")
            add(node["right_child"], name, "no")
    add(root)
    return graph
