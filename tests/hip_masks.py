"""Test helper: the dropout masks the HIP path applies, extracted AS DATA for the oracle.

Every dropout site of the HIP path multiplies by drop_scale(site_seed, flat index) where the flat index runs
over the site's logical row-major tensor (csrc/common.hpp).  Running the product kernel vqa_dropout on a tensor
of ones of that shape with the site's seed therefore yields the keep-scale mask of the site; the train-mode
parity tests hand these masks to oracle.vqa_forward(masks=...) -- if a fused kernel used another index
convention, or applied its mask at another place than the reference, the comparison fails."""
import torch


def hip_masks(engine, seed, B, T, g, device):
    from dl_vqa_amd import ops
    from dl_vqa_amd import engine as E

    def site(code, p, *shape):
        if p <= 0:
            return None
        ones = torch.ones(*shape, dtype=torch.float32, device=device)
        return ops.dropout(ones, p, E._site_seed(seed, code)).cpu()

    C, Q, mid, hid, Dc = engine.C, engine.Q, engine.mid, engine.hid, engine.Dc
    xld = 2 * mid if engine.do_option == "|" else mid
    nhwc = lambda t: None if t is None else t.permute(0, 3, 1, 2).contiguous()   # [B,g,g,ch] -> [B,ch,g,g]
    masks = {
        "image": nhwc(site(E.SITE_IMAGE, engine.p_image, B, g, g, C)),
        "text": site(E.SITE_TEXT, engine.p_text, B, T, engine.E),
        "att_v": nhwc(site(E.SITE_ATT_V, engine.p_att, B, g, g, C)),
        "att_q": site(E.SITE_ATT_Q, engine.p_att, B, Q),
        "att_x": nhwc(site(E.SITE_ATT_X, engine.p_att, B, g, g, xld)),
        "cls1": site(E.SITE_CLS1, engine.p_cls, B, Dc),
        "cls2": site(E.SITE_CLS2, engine.p_cls, B, hid),
    }
    torch.cuda.synchronize()
    return masks
