"""tree_map over dict / list / tuple pytrees with None and tensors as leaves (the subset of
lib/tree_util.py:56-76 that the policy API and behavioural_cloning.py:111 rely on)."""


def tree_map(f, tree):
    if isinstance(tree, dict):
        return type(tree)(**{k: tree_map(f, v) for k, v in tree.items()}) if type(tree) is not dict else {k: tree_map(f, v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(f, v) for v in tree)
    if tree is None:
        return None
    return f(tree)
