"""Which encoder kinds the loaded libmctsq_hip.so implements (forward / training)."""
FORWARD_KINDS = ("gru",)
TRAIN_KINDS = ("gru",)


def supported_kinds():
    return FORWARD_KINDS
