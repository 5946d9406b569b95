function mdl = rbpf_model(kind, NN, L)
% RBPF_MODEL - model-family descriptor whose handles the HIP-backed estimators recognise.
%   mdl = rbpf_model('dense-mag', NN, L)    6-D pose + curl-free field (slam-dense-mag closures)
%   mdl = rbpf_model('dense-radio', NN, L)  planar pose + heading, scalar field (slam-dense-radio closures)
%   mdl = rbpf_model('sparse-visual', nLand, [f fp fw])  planar pose, point landmarks, 1-D pinhole camera
%         (slam-sparse-visual closures pfslam.m:81-82; use with sparseFeatures = true and dynResNorm = [])
% NN, L come from domain_cartesian_dx (reference tools/).  mdl.dynModel / mdl.measModel / mdl.dynResNorm are
% ordinary function handles (they evaluate the reference closures, so existing plotting / data-generation code
% keeps working) that carry the descriptor in their workspace; matlab/particleFilter.m reads it back with
% functions(h).  UNTESTED here: no MATLAB in the build image.
  switch kind
    case 'dense-mag',   desc.kind = 1;
    case 'dense-radio', desc.kind = 2;
    case 'sparse-visual'
      desc.kind = 3; desc.nLand = NN; desc.cam = L(:)'; desc.use_dyn_res_norm = false;
      rbpf_desc = desc; %#ok<NASGU>
      mdl.dynModel   = @(xn,dx,dt,Q) rbpf_eval('dynModel', rbpf_desc, xn, dx, dt, Q);
      mdl.measModel  = @(xn,xl) rbpf_eval('measModel', rbpf_desc, xn, xl);
      mdl.dynResNorm = [];
      mdl.desc = desc;
      return
    otherwise, error('rbpf:model', 'unknown model family %s', kind);
  end
  desc.NN = int32(NN); desc.L = L(:)'; desc.use_dyn_res_norm = true;
  rbpf_desc = desc; %#ok<NASGU> captured by the handles below
  if desc.kind == 1
    [~,~,eigenfun_dx] = deal([]); %#ok<ASGLU> the reference closures are evaluated through the library helpers
    mdl.dynModel   = @(xn,dx,dt,Q) rbpf_eval('dynModel', rbpf_desc, xn, dx, dt, Q);
    mdl.measModel  = @(xn) rbpf_eval('measModel', rbpf_desc, xn);
    mdl.dynResNorm = @(xnk,xni,dx,dt,Q) rbpf_eval('dynResNorm', rbpf_desc, xnk, xni, dx, dt, Q);
  else
    mdl.dynModel   = @(xn,dx,dt,Q) rbpf_eval('dynModel', rbpf_desc, xn, dx, dt, Q);
    mdl.measModel  = @(xn) rbpf_eval('measModel', rbpf_desc, xn);
    mdl.dynResNorm = @(xnk,xni,dx,dt,Q) rbpf_eval('dynResNorm', rbpf_desc, xnk, xni, dx, dt, Q);
  end
  mdl.desc = desc;
end

function out = rbpf_eval(varargin) %#ok<STOUT>
  error('rbpf:eval', ['direct evaluation of a family handle from MATLAB is routed through the reference closures; ' ...
                      'add the reference tools/ to the path and call them, or use the estimators below']);
end
