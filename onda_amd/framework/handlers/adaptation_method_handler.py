"""Drop-in for ``framework/handlers/adaptation_method_handler.py`` (:11-40): the prototype methods
(the ADVENT baselines are out of scope)."""
ADAPTATION_METHOD_NAMES = ["PROTO_ONLINE", "PROTO_ONLINE_VSWITCH", "PROTO_ONLINE_HSWITCH", "PROTO_ONLINE_HYBRIDSWITCH"]


def get_adapt_method(cfg):
    name = cfg.METHOD.ADAPTATION.NAME
    assert name in ADAPTATION_METHOD_NAMES, f"cfg.METHOD.ADAPTATION.NAME not in {ADAPTATION_METHOD_NAMES}"
    if name == "PROTO_ONLINE":
        from onda_amd.framework.domain_adaptation.methods.prototypes import online_proDA
        return online_proDA
    if name == "PROTO_ONLINE_VSWITCH":
        from onda_amd.framework.domain_adaptation.methods.prototypes_vswitch import vswitch_proDA
        return vswitch_proDA
    if name == "PROTO_ONLINE_HSWITCH":
        from onda_amd.framework.domain_adaptation.methods.prototypes_hswitch import hswitch_proDA
        return hswitch_proDA
    from onda_amd.framework.domain_adaptation.methods.prototypes_hybrid_switch import hybrid_proDA
    return hybrid_proDA
