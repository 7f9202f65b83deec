"""Feature-column descriptors and the column map of the packed input matrix.

API mirror of deepctr/inputs.py:20-180 (SparseFeat / DenseFeat / VarLenSparseFeat,
get_feature_names, build_input_features, create_embedding_matrix, combined_dnn_input); the
descriptors are plain tuples with the reference's field names so user code that builds,
hashes, unpacks or `_replace`s them keeps working.
"""
import collections
from collections import OrderedDict

import torch
import torch.nn as nn

DEFAULT_GROUP_NAME = "default_group"

_SparseBase = collections.namedtuple(
    "SparseFeat", "name vocabulary_size embedding_dim use_hash dtype embedding_name group_name")
_DenseBase = collections.namedtuple("DenseFeat", "name dimension dtype")
_VarLenBase = collections.namedtuple("VarLenSparseFeat", "sparsefeat maxlen combiner length_name")


class SparseFeat(_SparseBase):
    """One categorical field: an id column of X looked up in a [vocabulary_size, embedding_dim] table."""
    __slots__ = ()

    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32", embedding_name=None,
                group_name=DEFAULT_GROUP_NAME):
        if embedding_dim == "auto":                     # deepctr/inputs.py:29-30
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        if use_hash:
            print("Notice! Feature Hashing on the fly currently is not supported in torch version,"
                  "you can use tensorflow version!")
        return _SparseBase.__new__(cls, name, vocabulary_size, embedding_dim, use_hash, dtype,
                                   name if embedding_name is None else embedding_name, group_name)

    def __hash__(self):
        return hash(self.name)


class DenseFeat(_DenseBase):
    """`dimension` consecutive float columns of X."""
    __slots__ = ()

    def __new__(cls, name, dimension=1, dtype="float32"):
        return _DenseBase.__new__(cls, name, dimension, dtype)

    def __hash__(self):
        return hash(self.name)


class VarLenSparseFeat(_VarLenBase):
    """Descriptor kept for API compatibility; the xDeepFM hot path has no variable-length fields
    (no config of the reference's scripts creates one) and the models reject it."""
    __slots__ = ()

    def __new__(cls, sparsefeat, maxlen, combiner="mean", length_name=None):
        return _VarLenBase.__new__(cls, sparsefeat, maxlen, combiner, length_name)

    name = property(lambda self: self.sparsefeat.name)
    vocabulary_size = property(lambda self: self.sparsefeat.vocabulary_size)
    embedding_dim = property(lambda self: self.sparsefeat.embedding_dim)
    use_hash = property(lambda self: self.sparsefeat.use_hash)
    dtype = property(lambda self: self.sparsefeat.dtype)
    embedding_name = property(lambda self: self.sparsefeat.embedding_name)
    group_name = property(lambda self: self.sparsefeat.group_name)

    def __hash__(self):
        return hash(self.name)


def build_input_features(feature_columns):
    """OrderedDict name -> (first column, one-past-last column) of the packed matrix X; a name seen
    twice keeps its first slot (deepctr/inputs.py:99-123)."""
    index = OrderedDict()
    cursor = 0
    for fc in feature_columns:
        if not isinstance(fc, (SparseFeat, DenseFeat, VarLenSparseFeat)):
            raise TypeError("Invalid feature column type,got", type(fc))
        if fc.name in index:
            continue
        if isinstance(fc, SparseFeat):
            width = 1
        elif isinstance(fc, DenseFeat):
            width = fc.dimension
        else:
            width = fc.maxlen
        index[fc.name] = (cursor, cursor + width)
        cursor += width
        if isinstance(fc, VarLenSparseFeat) and fc.length_name is not None and fc.length_name not in index:
            index[fc.length_name] = (cursor, cursor + 1)
            cursor += 1
    return index


def get_feature_names(feature_columns):
    return list(build_input_features(feature_columns).keys())


def split_columns(feature_columns):
    cols = list(feature_columns) if feature_columns else []
    sparse = [c for c in cols if isinstance(c, SparseFeat)]
    dense = [c for c in cols if isinstance(c, DenseFeat)]
    varlen = [c for c in cols if isinstance(c, VarLenSparseFeat)]
    return sparse, dense, varlen


def create_embedding_matrix(feature_columns, init_std=0.0001, linear=False, sparse=False, device="cpu"):
    """nn.ModuleDict {embedding_name: nn.Embedding}.  Tables are first all constructed (default
    N(0,1) draw) and then all re-drawn N(0, init_std), the RNG order of deepctr/inputs.py:167-178,
    so a given torch seed yields the reference's initial tables."""
    sparse_cols, _, varlen = split_columns(feature_columns)
    tables = nn.ModuleDict()
    for fc in sparse_cols + varlen:
        tables[fc.embedding_name] = nn.Embedding(fc.vocabulary_size, 1 if linear else fc.embedding_dim, sparse=sparse)
    for emb in tables.values():
        nn.init.normal_(emb.weight, mean=0, std=init_std)
    return tables.to(device)


def combined_dnn_input(sparse_embedding_list, dense_value_list):
    """Same contract as deepctr/inputs.py:126-138 for callers that hold [B,1,D] / [B,k] lists."""
    parts = []
    if len(sparse_embedding_list) > 0:
        parts.append(torch.flatten(torch.cat(sparse_embedding_list, dim=-1), start_dim=1))
    if len(dense_value_list) > 0:
        parts.append(torch.flatten(torch.cat(dense_value_list, dim=-1), start_dim=1))
    if not parts:
        raise NotImplementedError
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
