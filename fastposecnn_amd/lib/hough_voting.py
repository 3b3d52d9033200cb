"""Drop-in for HoughVotingLayer.forward (F/lib/hough_voting.py:33-63).
Lines :68-579 of the reference file are an older pure-torch voting scheme that is commented out
of forward (:48) and unreachable; it is not provided."""
import torch
import torch.nn as nn

import ransac_voting_gpu_layer.ransac_voting_gpu as rvg


class HoughVotingLayer(nn.Module):

    def __init__(self, HPARAM):
        super().__init__()
        self.HPARAM = HPARAM

    def forward(self, agg_data, n_dev=None, seed=None, inv_intrinsics=None):
        uv_img = agg_data['xy']                # [n,2,H,W] masked vote field
        mask = agg_data['instance_masks']      # [n,H,W]
        # [n,H,W,1,2] strided VIEW of the two planes — read in place by the kernel
        reshaped_uv_img = torch.unsqueeze(uv_img.permute(0, 2, 3, 1), dim=3)
        from aggregation_layer import mask_bits_of
        # (not in the reference) with the inverse intrinsics given, the RT assembly that pose_regressor.py runs next
        # (gtf.samplewise_get_RT) rides on the vote's last kernel: one launch less per frame
        pose = None
        if inv_intrinsics is not None and mask.shape[0] > 0 and not torch.is_grad_enabled():
            n = mask.shape[0]
            f32 = dict(dtype=torch.float32, device=mask.device)
            pose = dict(q=agg_data['quaternion'].to(torch.float32).contiguous(), z=agg_data['z'].to(torch.float32).reshape(-1).contiguous(),
                        kinv=inv_intrinsics.to(**f32).contiguous(), R=torch.empty((n, 3, 3), **f32), T=torch.empty((n, 3), **f32),
                        RT=torch.empty((n, 4, 4), **f32))
        output = rvg.ransac_voting_layer_v3(
            mask=mask,
            vertex=reshaped_uv_img,
            round_hyp_num=self.HPARAM.HV_NUM_OF_HYPOTHESES,
            n_dev=n_dev,
            seed=seed,
            mask_bits=mask_bits_of(mask),      # set when `mask` is the aggregation layer's own output: the scan skips the f32 planes
            pose=pose,
        )
        good_output = torch.squeeze(output, dim=1)
        agg_data.update({'hypothesis': output, 'pruned_hypothesis': output, 'xy': good_output, 'xy_mask': uv_img})
        if pose is not None:
            agg_data.update({'R': pose['R'], 'T': pose['T'], 'RT': pose['RT']})
        return agg_data
