"""Which encoder kinds the loaded libmctsq_hip.so implements (forward / training)."""
FORWARD_KINDS = ("gru", "conv", "mlp", "transformer")
TRAIN_KINDS = ("gru", "conv", "mlp", "transformer")


def supported_kinds():
    return FORWARD_KINDS
