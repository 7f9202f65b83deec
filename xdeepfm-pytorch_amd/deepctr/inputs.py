from xdfm_amd.inputs import (DEFAULT_GROUP_NAME, DenseFeat, SparseFeat, VarLenSparseFeat,  # noqa: F401
                             build_input_features, combined_dnn_input, create_embedding_matrix,
                             get_feature_names)
