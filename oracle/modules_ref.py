"""TEST INFRASTRUCTURE ONLY -- CPU restatement (stock torch fp32 ops) of the reference's hot-path modules.

Functional style over a reference-layout state_dict (`sd`: key -> tensor), with every random draw injected
as an argument, so that the same inputs/weights/noise can be pushed through (a) the real reference
(tests/golden/make_golden.py, build container only), (b) this restatement and (c) the HIP modules.
Pinned by tests/test_oracle_golden.py against the committed golden vectors.  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this; the product never routes through it.

Each function cites the reference lines it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ops as O


def sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _wnw(sd, name):
    """weight of a (possibly weight-normalised) conv: reference keys weight | weight_g + weight_v."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    return O.weight_norm(sd[name + ".weight_v"], sd[name + ".weight_g"])


def _bn(x, sd, name, training, eps=1e-5):
    """nn.BatchNorm1d forward: batch statistics in training mode (running stats are not updated here)."""
    if training:
        return F.batch_norm(x, None, None, sd[name + ".weight"], sd[name + ".bias"], True, 0.1, eps)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], False, 0.1, eps)


# ----------------------------------------------------------------------------------------------------------
# WN / FVAE  (modules/fastspeech/fs2_vae.py, modules/voice_conversion/vae_models.py)
# ----------------------------------------------------------------------------------------------------------
def wn_forward(sd, x, x_mask, g, hidden, kernel_size, n_layers, dilation_rate=1):
    """WN.forward, fs2_vae.py:61-91 (p_dropout = 0).  x [B,C,T], x_mask [B,1,T], g [B,gin,T]."""
    out = torch.zeros_like(x)
    if g is not None:
        g = F.conv1d(g, _wnw(sd, "cond_layer"), sd["cond_layer.bias"])                          # :70-71
    for i in range(n_layers):
        dil = dilation_rate ** i
        pad = int((kernel_size * dil - dil) / 2)                                               # :45-46
        x_in = F.conv1d(x, _wnw(sd, f"in_layers.{i}"), sd[f"in_layers.{i}.bias"], 1, pad, dil)  # :74
        g_l = g[:, i * 2 * hidden:(i + 1) * 2 * hidden] if g is not None else torch.zeros_like(x_in)  # :76-80
        acts = O.wn_gate(x_in, g_l)                                                             # :82
        rs = F.conv1d(acts, _wnw(sd, f"res_skip_layers.{i}"), sd[f"res_skip_layers.{i}.bias"])  # :84
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask                                                   # :86
            out = out + rs[:, hidden:]                                                          # :87
        else:
            out = out + rs                                                                      # :89
    return out * x_mask                                                                         # :90


def global_encoder(sd, x, x_mask, g_sqz, eps, hp, training):
    """GlobalFVAEEncoder.forward, vae_models.py:96-105.  eps = the randn_like draw (injected)."""
    H, L = hp["fvae_enc_dec_hidden"], hp["latent_size"]
    x = F.conv1d(x, sd["pre_net.0.weight"], sd["pre_net.0.bias"], 4, 2)                          # fs2_vae.py:109-114
    m = x_mask[:, :, ::4][:, :, :x.shape[-1]]                                                   # :98
    x = x * m
    x = wn_forward(sub(sd, "wn."), x, m, g_sqz, H, hp["fvae_kernel_size"], hp["fvae_enc_n_layers"]) * m
    x = F.conv1d(x, sd["out_proj.weight"], sd["out_proj.bias"])                                 # :101
    p = x
    for i, last in ((0, False), (3, False), (6, True)):                                         # :86-94
        p = F.conv1d(p, sd[f"poolings.{i}.weight"], sd[f"poolings.{i}.bias"], 2)
        if not last:
            p = _bn(F.relu(p), sd, f"poolings.{i + 2}", training)
    x = p.mean(-1, keepdim=True)                                                                # :102
    m_q, logs_q = torch.split(x, L, dim=1)                                                      # :103
    z = m_q + eps * torch.exp(logs_q)                                                           # :104
    return z, m_q, logs_q, m


def global_decoder(sd, z, x_mask, g, hp):
    """GlobalFVAEDecoder.forward, vae_models.py:124-131."""
    H = hp["fvae_enc_dec_hidden"]
    x = z.repeat(1, 1, g.shape[-1] // 4)                                                        # :126
    x = F.conv_transpose1d(x, sd["pre_net.0.weight"], sd["pre_net.0.bias"], 4)                  # :127
    x = x * x_mask
    x = wn_forward(sub(sd, "wn."), x, x_mask, g, H, hp["fvae_kernel_size"], hp["fvae_dec_n_layers"]) * x_mask
    return F.conv1d(x, sd["out_proj.weight"], sd["out_proj.bias"])                              # :130


def global_fvae(sd, x, x_mask, g, eps, hp, training):
    """TMPFVAE.forward (infer=False), vae_models.py:12-41, use_prior_glow=False."""
    g_sqz = F.conv1d(g, sd["g_pre_net.0.weight"], sd["g_pre_net.0.bias"], 4, 2)                 # :20, fs2_vae.py:164-167
    z_q, m_q, logs_q, mask_sqz = global_encoder(sub(sd, "encoder."), x, x_mask, g_sqz, eps, hp, training)
    x_recon = global_decoder(sub(sd, "decoder."), z_q, x_mask, g, hp)                           # :23
    # KL(N(m, e^logs) || N(0,1)) per element (torch.distributions.kl_divergence, :38)
    var = torch.exp(logs_q) ** 2
    kl = 0.5 * (var + m_q ** 2 - 1.0) - logs_q
    loss_kl = (kl * mask_sqz).sum() / mask_sqz.sum() / z_q.shape[1]                             # :39
    return x_recon, loss_kl, m_q, logs_q, mask_sqz, z_q


# ----------------------------------------------------------------------------------------------------------
# conditioning path  (modules/voice_conversion/svb_vae.py:60-86, common_layers.py:672-773)
# ----------------------------------------------------------------------------------------------------------
def conv_stacks(sd, x, n_layers=3):
    """ConvStacks.forward (norm='gn', res=True), common_layers.py:688-707,761-773.  x [B,T,H]."""
    x = F.linear(x, sd["in_proj.weight"], sd["in_proj.bias"]).transpose(1, -1)
    for i in range(n_layers):
        w = sd[f"conv.{i}.conv.conv.weight"]
        h = F.conv1d(x, w, sd[f"conv.{i}.conv.conv.bias"], 1, (w.shape[-1] - 1) // 2)            # ConvNorm 'same' padding
        h = F.group_norm(h, w.shape[0] // 16, sd[f"conv.{i}.norm.weight"], sd[f"conv.{i}.norm.bias"])
        x = x + F.relu(h)
    return F.linear(x.transpose(1, -1), sd["out_proj.weight"], sd["out_proj.bias"])


def rel_pos_encoding(T, d_model):
    """RelPositionalEncoding (reverse=True), espnet_positional_embedding.py:23-46,107-112 -> pe[:, :T] of a 5000-long table.

    The table is built once for max_len=5000 in reversed order (position 4999 first), and the forward takes the FIRST T
    rows of it (espnet_positional_embedding.py:111)."""
    max_len = 5000
    position = torch.arange(max_len - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe[:T].unsqueeze(0)


def rel_mha(sd, x, pos_emb, mask, n_head=4):
    """RelPositionMultiHeadedAttention.forward, espnet_transformer_attn.py:150-186 (+ forward_attention :61-89).
    mask [B,1,T] bool (True = keep)."""
    B, T, D = x.shape
    dk = D // n_head
    q = F.linear(x, sd["linear_q.weight"], sd["linear_q.bias"]).view(B, T, n_head, dk)
    k = F.linear(x, sd["linear_k.weight"], sd["linear_k.bias"]).view(B, T, n_head, dk).transpose(1, 2)
    v = F.linear(x, sd["linear_v.weight"], sd["linear_v.bias"]).view(B, T, n_head, dk).transpose(1, 2)
    p = F.linear(pos_emb, sd["linear_pos.weight"]).view(1, -1, n_head, dk).transpose(1, 2)
    q_u = (q + sd["pos_bias_u"]).transpose(1, 2)
    q_v = (q + sd["pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(q_u, k.transpose(-2, -1))
    bd = torch.matmul(q_v, p.transpose(-2, -1))
    # rel_shift, :125-148
    zero_pad = torch.zeros((*bd.size()[:3], 1))
    bd_p = torch.cat([zero_pad, bd], dim=-1).view(B, n_head, T + 1, T)
    bd = bd_p[:, :, 1:].view_as(bd)
    scores = (ac + bd) / math.sqrt(dk)
    mm = mask.unsqueeze(1).eq(0)
    scores = scores.masked_fill(mm, float(np.finfo(np.float32).min))
    attn = torch.softmax(scores, dim=-1).masked_fill(mm, 0.0)
    o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, D)
    return F.linear(o, sd["linear_out.weight"], sd["linear_out.bias"])


def conformer_layer(sd, x, pos_emb, mask):
    """EncoderLayer.forward (macaron, normalize_before, eval/no dropout), conformer/layers.py:182-258."""
    def ln(name, t):
        return F.layer_norm(t, (t.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)

    def ffn(name, t):  # MultiLayeredConv1d (kernel 1), layers.py:112-121
        h = torch.relu(F.conv1d(t.transpose(-1, 1), sd[name + ".w_1.weight"], sd[name + ".w_1.bias"]))
        return F.conv1d(h, sd[name + ".w_2.weight"], sd[name + ".w_2.bias"]).transpose(-1, 1)

    x = x + 0.5 * ffn("feed_forward_macaron", ln("norm_ff_macaron", x))                        # :200-206
    x = x + rel_mha(sub(sd, "self_attn."), ln("norm_mha", x), pos_emb, mask)                    # :209-230
    h = ln("norm_conv", x).transpose(1, 2)                                                      # :235-241, ConvolutionModule :48-68
    h = F.glu(F.conv1d(h, sd["conv_module.pointwise_conv1.weight"], sd["conv_module.pointwise_conv1.bias"]), dim=1)
    dw = sd["conv_module.depthwise_conv.weight"]
    h = F.conv1d(h, dw, sd["conv_module.depthwise_conv.bias"], 1, (dw.shape[-1] - 1) // 2, 1, dw.shape[0])
    h = _bn(h, sd, "conv_module.norm", False)
    h = h * torch.sigmoid(h)
    h = F.conv1d(h, sd["conv_module.pointwise_conv2.weight"], sd["conv_module.pointwise_conv2.bias"]).transpose(1, 2)
    x = x + h
    x = x + 0.5 * ffn("feed_forward", ln("norm_ff", x))                                        # :244-249
    return ln("norm_final", x)                                                                  # :251-252


def vc_asr_content(sd, mel, hp):
    """VCASR.forward(prev_tokens=None) in eval mode: Prenet (pe.py:23-41) + ConformerLayers (conformer.py:37-53)."""
    strides = hp["mel_strides"]
    padding_mask = mel.abs().sum(-1).eq(0)
    nonpad = 1 - padding_mask.float()[:, None, :]
    x = mel.transpose(1, 2)
    for i, s in enumerate(strides):
        nonpad = nonpad[:, :, ::s]
        w = sd[f"mel_prenet.layers.{i}.0.weight"]
        x = F.conv1d(x, w, sd[f"mel_prenet.layers.{i}.0.bias"], s, w.shape[-1] // 2)
        x = _bn(F.relu(x), sd, f"mel_prenet.layers.{i}.2", False) * nonpad
    x = F.linear(x.transpose(1, 2), sd["mel_prenet.out_proj.weight"], sd["mel_prenet.out_proj.bias"])
    x = x * nonpad.transpose(1, 2)
    # ConformerLayers
    ce = sub(sd, "content_encoder.")
    nonpadding_mask = x.abs().sum(-1) > 0
    D = x.shape[-1]
    pos_emb = rel_pos_encoding(x.shape[1], D).to(x.dtype)     # (fp32 table as the reference builds it; cast for the fp64 arbiter)
    x = x * math.sqrt(D)
    for li in range(hp["asr_enc_layers"]):
        x = conformer_layer(sub(ce, f"encoder_layers.{li}."), x, pos_emb, nonpadding_mask[:, None, :])
    if hp.get("asr_last_norm", False):
        x = F.layer_norm(x, (D,), ce["layer_norm.weight"], ce["layer_norm.bias"])
    else:
        x = F.linear(x, ce["layer_norm.weight"], ce["layer_norm.bias"])                         # conformer.py:30-33
    return x * nonpadding_mask.float()[:, :, None]


def upsample_layer(sd, h, hp, training):
    """svb_vae.py:39-45 with asr_upsample_norm == 'bn'.  h [B,H,T/2] -> [B,H,T]."""
    idx = 0
    for scale in hp["mel_strides"]:
        if scale > 1:
            h = F.interpolate(h, scale_factor=scale, mode="nearest")
            h = F.conv1d(h, sd[f"upsample_layer.{idx}.1.weight"], sd[f"upsample_layer.{idx}.1.bias"], 1, scale)
            h = _bn(F.relu(h), sd, f"upsample_layer.{idx}.3", training)
            idx += 1
    return F.conv1d(h, sd[f"upsample_layer.{idx}.weight"], sd[f"upsample_layer.{idx}.bias"], 1, 2)


def prepare_condition(sd, mel, pitch, spk, hp, training):
    """SVBVAE.prepare_condition, svb_vae.py:60-86."""
    T = pitch.shape[1]
    h_pitch = conv_stacks(sub(sd, "pitch_encoder."), F.embedding(pitch, sd["pitch_embed.weight"], 0))
    h_content = vc_asr_content(sub(sd, "vc_asr."), mel, hp).detach()
    h_content = upsample_layer(sd, h_content.transpose(1, 2), hp, training).transpose(1, 2)[:, :mel.shape[1]]
    h_style = F.linear(spk, sd["spk_embed_proj.weight"], sd["spk_embed_proj.bias"])[:, None, :].repeat(1, T, 1)
    return dict(h_pitch=h_pitch, h_content=h_content, h_style=h_style, tgt_nonpadding=(pitch > 0).float()[:, :, None])


def normal_vae(sd, tgt_mel, c, eps, hp, training):
    """SVBVAE.normal_vae, svb_vae.py:152-162."""
    cond = F.linear(torch.cat([c["h_pitch"], c["h_content"], c["h_style"]], -1), sd["encoded_embed_proj.weight"],
                    sd["encoded_embed_proj.bias"]).transpose(1, 2)
    mel_out, kl, m_q, logs_q, mask_sqz, z_q = global_fvae(sub(sd, "vae_model."), tgt_mel.transpose(1, 2),
                                                          c["tgt_nonpadding"].transpose(1, 2), cond, eps, hp, training)
    return dict(mel_out=mel_out.transpose(1, 2), kl=kl, m_q=m_q, logs_q=logs_q, x_mask_sqz=mask_sqz, z_q=z_q)


def global_latent_map(sd, z, spk_emb, training):
    """GlobalLatentMap.forward, vae_models.py:167-172."""
    s = spk_emb[:, :, :z.shape[-1]]
    s = F.conv1d(F.relu(F.conv1d(s, sd["spk_proj.0.weight"], sd["spk_proj.0.bias"])), sd["spk_proj.2.weight"],
                 sd["spk_proj.2.bias"])
    x = z + s
    x = F.relu(_bn(F.conv1d(x, sd["convs.0.weight"], sd["convs.0.bias"]), sd, "convs.1", training))
    x = F.relu(_bn(F.conv1d(x, sd["convs.3.weight"], sd["convs.3.bias"]), sd, "convs.4", training))
    return F.conv1d(x, sd["convs.6.weight"], sd["convs.6.bias"])


def mle_svb_vae(sd, mels, prof_mels, pitch, prof_pitch, spk, a2p_alignment, ways, eps_a2a, eps_p2p, hp, training=True,
                map_training=None):
    """MleSVBVAE.forward, svb_vae.py:258-312 (both speaker ids = the amateur embedding, svb_vae_task.py:145)."""
    if map_training is None:
        map_training = training
    ca = prepare_condition(sd, mels, pitch, spk, hp, training)
    cp = prepare_condition(sd, prof_mels, prof_pitch, spk, hp, training)
    ret = {}
    if "a2a" in ways:
        ret["a2a"] = normal_vae(sd, mels, ca, eps_a2a, hp, training)
    if "p2p" in ways:
        ret["p2p"] = normal_vae(sd, prof_mels, cp, eps_p2p, hp, training)
    if "a2p" in ways:
        z_a = ret["a2a"]["z_q"]
        m_p, logs_p = ret["p2p"]["m_q"], ret["p2p"]["logs_q"]
        z_map = global_latent_map(sub(sd, "z_mapping_function."), z_a, ca["h_style"].transpose(1, 2), map_training)
        sigma = logs_p.exp()
        log_prob = -((z_map - m_p) ** 2) / (2 * sigma ** 2) - sigma.log() - math.log(math.sqrt(2 * math.pi))
        mle = -log_prob.sum() / z_map.shape[0] / z_map.shape[1]                                 # :293
        idx = a2p_alignment[:, :, None].repeat(1, 1, hp["hidden_size"])                         # :285
        cond = F.linear(torch.cat([cp["h_pitch"], torch.gather(ca["h_content"], 1, idx),
                                   ca["h_style"][:, :1, :].repeat(1, cp["h_pitch"].shape[1], 1)], -1),
                        sd["encoded_embed_proj.weight"], sd["encoded_embed_proj.bias"]).transpose(1, 2)  # :296-300
        mel = global_decoder(sub(sd, "vae_model.decoder."), z_map, cp["tgt_nonpadding"].transpose(1, 2), cond, hp)
        ret["a2p"] = dict(mle=mle, mel_out=mel.transpose(1, 2))
    return ret, ca, cp


# ----------------------------------------------------------------------------------------------------------
# losses  (tasks/tts/fs2.py:143-175, modules/commons/ssim.py:320-351, tasks/tts/tts.py:127-131)
# ----------------------------------------------------------------------------------------------------------
def weights_nonzero_speech(target):
    return target.abs().sum(-1, keepdim=True).ne(0).float().repeat(1, 1, target.size(-1))


def l1_loss(out, target):
    w = weights_nonzero_speech(target)
    return (F.l1_loss(out, target, reduction="none") * w).sum() / w.sum()


def ssim_map(img1, img2, window_size=11):
    """_ssim(size_average=False) on [B,1,T,80]: returns ssim_map.mean(1) -> [B,T,80].  ssim.py:320-351."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    p = window_size // 2
    mu1, mu2 = F.conv2d(img1, window, padding=p), F.conv2d(img2, window, padding=p)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=p) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=p) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=p) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean(1)


def ssim_loss(out, target, bias=6.0):
    w = weights_nonzero_speech(target)
    s = 1 - ssim_map(out[:, None] + bias, target[:, None] + bias)
    return (s * w).sum() / w.sum()


# ----------------------------------------------------------------------------------------------------------
# mel discriminator  (modules/fastspeech/multi_window_disc.py)
# ----------------------------------------------------------------------------------------------------------
def mel_discriminator(sd, x, starts, win_lengths=(32, 64, 128), drop_masks=None):
    """Discriminator.forward (uncond, norm 'in', reduction 'stack'), multi_window_disc.py:180-199 with the window
    starts injected (:144-148).  drop_masks: optional per-(window, block) channel keep-masks [B,C] already scaled
    by 1/(1-p) (Dropout2d, :23); None = eval mode."""
    if x.dim() == 3:
        x = x[:, None]
    ys, hs = [], []
    for wi, wl in enumerate(win_lengths):
        s = int(starts[wi][0])
        h = x[:, :, s:s + wl]
        for bi in range(3):
            pre = f"discriminator.conv_layers.{wi}.model.{bi}."
            h = F.leaky_relu(F.conv2d(h, sd[pre + "0.weight"], sd[pre + "0.bias"], 2, 1), 0.2)
            if drop_masks is not None:
                h = h * drop_masks[wi][bi][:, :, None, None]
            if bi > 0:
                h = F.instance_norm(h, weight=sd[pre + "3.weight"], bias=sd[pre + "3.bias"], eps=1e-5)
            hs.append(h)
        al = f"discriminator.conv_layers.{wi}.adv_layer."
        ys.append(F.linear(h.reshape(h.shape[0], -1), sd[al + "weight"], sd[al + "bias"]))
    return torch.stack(ys, -1), hs


# ----------------------------------------------------------------------------------------------------------
# NSF-HifiGAN  (modules/hifigan/hifigan.py, modules/parallel_wavegan/models/source.py)
# ----------------------------------------------------------------------------------------------------------
def hifigan_generator(sd, mel, f0, rand_ini, noise, cfg, lrelu=0.1):
    """HifiGanGenerator.forward, hifigan.py:144-169 (ResBlock1 :54-61)."""
    rates, ksz = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    rk, rd = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    upp = int(np.prod(rates))
    har, _, _ = O.sine_source(f0, rand_ini, noise, sd["m_source.l_linear.weight"].view(-1),
                              sd["m_source.l_linear.bias"], upp, float(cfg["audio_sample_rate"]))
    har = har[:, None, :]                                                                       # [B,1,L]
    x = F.conv1d(mel, _wnw(sd, "conv_pre"), sd["conv_pre.bias"], 1, 3)
    nk = len(rk)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, lrelu)
        x = F.conv_transpose1d(x, _wnw(sd, f"ups.{i}"), sd[f"ups.{i}.bias"], u, (k - u) // 2)
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            xs = F.conv1d(har, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"], s, s // 2)
        else:
            xs = F.conv1d(har, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"])
        x = x + xs
        acc = None
        for j in range(nk):
            r = x
            pre = f"resblocks.{i * nk + j}."
            for m, d in enumerate(rd[j]):
                t = F.leaky_relu(r, lrelu)
                t = F.conv1d(t, _wnw(sd, pre + f"convs1.{m}"), sd[pre + f"convs1.{m}.bias"], 1, (rk[j] * d - d) // 2, d)
                t = F.leaky_relu(t, lrelu)
                t = F.conv1d(t, _wnw(sd, pre + f"convs2.{m}"), sd[pre + f"convs2.{m}.bias"], 1, (rk[j] - 1) // 2)
                r = t + r
            acc = r if acc is None else acc + r
        x = acc / nk
    x = F.leaky_relu(x)                                                                         # default slope 0.01, :165
    x = F.conv1d(x, _wnw(sd, "conv_post"), sd["conv_post.bias"], 1, 3)
    return torch.tanh(x)


def _spectral_w(sd, name, train=False):
    """spectral_norm: W / sigma with sigma = u^T W_mat v from the stored buffers; in train mode one power iteration
    (v = normalize(W^T u), u = normalize(W v), buffers updated in place) runs first, as torch.nn.utils.spectral_norm does on
    every training forward (reference hifigan.py:261,294-296)."""
    w = sd[name + ".weight_orig"]
    wm = w.flatten(1)
    if train:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), sd[name + ".weight_u"]), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
            sd[name + ".weight_v"].copy_(v)
            sd[name + ".weight_u"].copy_(u)
    sigma = torch.dot(sd[name + ".weight_u"].clone(), torch.mv(wm, sd[name + ".weight_v"].clone()))
    return w / sigma


def disc_period(sd, x, period, lrelu=0.1):
    """DiscriminatorP.forward, hifigan.py:202-223."""
    fmap = []
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    for i, stride in enumerate((3, 3, 3, 3, 1)):
        x = F.leaky_relu(F.conv2d(x, _wnw(sd, f"convs.{i}"), sd[f"convs.{i}.bias"], (stride, 1), (2, 0)), lrelu)
        fmap.append(x)
    x = F.conv2d(x, _wnw(sd, "conv_post"), sd["conv_post.bias"], 1, (1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_scale(sd, x, spectral, lrelu=0.1, train=False):
    """DiscriminatorS.forward, hifigan.py:273-286."""
    cfgs = [(1, 7, 1), (2, 20, 4), (2, 20, 16), (4, 20, 16), (4, 20, 16), (1, 20, 16), (1, 2, 1)]
    wf = (lambda n: _spectral_w(sd, n, train)) if spectral else (lambda n: _wnw(sd, n))
    fmap = []
    for i, (s, p, g) in enumerate(cfgs):
        x = F.leaky_relu(F.conv1d(x, wf(f"convs.{i}"), sd[f"convs.{i}.bias"], s, p, 1, g), lrelu)
        fmap.append(x)
    x = F.conv1d(x, wf("conv_post"), sd["conv_post.bias"], 1, 1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def multi_period_disc(sd, y, y_hat, periods=(2, 3, 5, 7, 11)):
    """MultiPeriodDiscriminator.forward, hifigan.py:237-250."""
    outs = ([], [], [], [])
    for i, p in enumerate(periods):
        d = sub(sd, f"discriminators.{i}.")
        r, fr = disc_period(d, y, p)
        g, fg = disc_period(d, y_hat, p)
        for o, v in zip(outs, (r, g, fr, fg)):
            o.append(v)
    return outs


def multi_scale_disc(sd, y, y_hat, train=False):
    """MultiScaleDiscriminator.forward, hifigan.py:309-325 (scale 0 spectral-norm; train: power iteration per call)."""
    outs = ([], [], [], [])
    for i in range(3):
        if i != 0:
            y, y_hat = F.avg_pool1d(y, 4, 2, 1), F.avg_pool1d(y_hat, 4, 2, 1)
        d = sub(sd, f"discriminators.{i}.")
        r, fr = disc_scale(d, y, i == 0, train=train)
        g, fg = disc_scale(d, y_hat, i == 0, train=train)
        for o, v in zip(outs, (r, g, fr, fg)):
            o.append(v)
    return outs


def feature_loss(fmap_r, fmap_g):
    """hifigan.py:328-334."""
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl - gl))
    return loss * 2


def discriminator_loss(dr_list, dg_list):
    """hifigan.py:337-347."""
    r = sum(torch.mean((1 - dr) ** 2) for dr in dr_list) / len(dr_list)
    g = sum(torch.mean(dg ** 2) for dg in dg_list) / len(dr_list)
    return r, g


def generator_loss(dg_list):
    """hifigan.py:359-365."""
    return sum(torch.mean((1 - dg) ** 2) for dg in dg_list) / len(dg_list)
