"""Drop-in for ``framework/handlers/adaptation_method_handler.py`` (:11-40), prototype methods only."""
ADAPTATION_METHOD_NAMES = ["PROTO_ONLINE", "PROTO_ONLINE_HYBRIDSWITCH"]


def get_adapt_method(cfg):
    name = cfg.METHOD.ADAPTATION.NAME
    assert name in ADAPTATION_METHOD_NAMES, f"cfg.METHOD.ADAPTATION.NAME not in {ADAPTATION_METHOD_NAMES}"
    if name == "PROTO_ONLINE":
        from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA
        return online_proDA
    from onda_amd.framework.domain_adaptation.methods.prototypes_hybrid_switch import hybrid_proDA
    return hybrid_proDA
