function mdl = rbpf_model(kind, NN, L)
% RBPF_MODEL - explicit model-family descriptor (optional: the wrappers recognise the reference's own handles).
%   mdl = rbpf_model('dense-mag', NN, L)    6-D pose + curl-free field (slam-dense-mag closures)
%   mdl = rbpf_model('dense-radio', NN, L)  planar pose + heading, scalar field (slam-dense-radio closures)
%   mdl = rbpf_model('sparse-visual', nLand, [f fp fw])  planar pose, point landmarks, 1-D pinhole camera
%         (slam-sparse-visual closures pfslam.m:81-82; use with sparseFeatures = true and dynResNorm = [])
% NN, L come from domain_cartesian_dx (reference tools/).  mdl.dynModel / mdl.measModel / mdl.dynResNorm are ordinary
% function handles with the reference's calling conventions; called directly they evaluate the family's closures on the
% device through the MEX gateway (dynModel draws its randn's from MATLAB's stream first, in the closures' order), so
% plotting and data-generation code keeps working.  They carry the descriptor in their workspace, which the estimators
% read back with functions(h).  UNTESTED under MATLAB here: no MATLAB in the build image.
  switch kind
    case 'dense-mag',   desc.kind = 1; nw = 6;
    case 'dense-radio', desc.kind = 2; nw = 1;
    case 'sparse-visual'
      desc.kind = 3; desc.nLand = NN; desc.cam = L(:)'; desc.use_dyn_res_norm = false;
      rbpf_desc = desc; %#ok<NASGU> captured by the handles below
      f = L(1); fp = L(2); fw = L(3);
      mdl.dynModel   = @(xn,dx,dt,Q) rbpf_keep(rbpf_desc, xn + dx' + sqrt(dt*Q)*randn(size(xn,1),1));       % pfslam.m:81
      mdl.measModel  = @(xn,xl) rbpf_keep2(rbpf_desc, xn, xl, f, fp, fw);                                   % pfslam.m:82
      mdl.dynResNorm = [];
      mdl.desc = desc;
      return
    otherwise, error('rbpf:model', 'unknown model family %s', kind);
  end
  desc.NN = int32(NN); desc.L = L(:)'; desc.use_dyn_res_norm = true;
  rbpf_desc = desc;
  mdl.dynModel   = @(xn,dx,dt,Q) rbpf_mex('dynModel', rbpf_desc, xn, dx, dt, Q, randn(nw, size(xn,2)));
  mdl.measModel  = @(xn) rbpf_mex('measModel', rbpf_desc, xn);
  mdl.dynResNorm = @(xnk,xni,dx,dt,Q) rbpf_mex('dynResNorm', rbpf_desc, xnk, xni, dx, dt, Q);
  mdl.desc = desc;
end

function v = rbpf_keep(~, v)
end

function [yhat, dy] = rbpf_keep2(~, xn, xl, f, fp, fw)
  [yhat, dy] = measurement([xn(1:3); xl], f, fp, fw, true);                  % the reference's own function (slam-sparse-visual)
end
