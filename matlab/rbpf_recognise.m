function desc = rbpf_recognise(dynModel, measModel, dynResNorm, sparseFeatures)
% RBPF_RECOGNISE - map the reference's OWN model handles to a model family of the HIP library.
%
% The examples of the reference pass plain MATLAB handles: nested functions (@dynModel / @measModel of
% examples/slam-dense-mag/run_dense3D_magfield.m:142-144,265-279,301-308) and anonymous functions
% (examples/slam-dense-radio/run_dense2D_withHeading.m:75-77,168; examples/slam-sparse-visual/pfslam.m:81-82).  A HIP
% kernel cannot call them, so they are recognised and replaced by the kernels' own implementation of the same closures:
%   1. PROPOSE a family from functions(h): the nested function's name / file, or the text of the anonymous function, and
%      pull the constants the closure captured (NN, LL or the eigenfunction handle's L, camera f fp fw) from its workspace;
%   2. VERIFY the proposal by evaluating the handle once next to the library's implementation of that family
%      (rbpf_mex('measModel' | 'dynModel' | 'dynResNorm', ...)) -- dynModel with MATLAB's generator rewound, so that the same
%      randn values feed both -- and accept only on agreement to 1e-10.
% Anything else (edited closures, user models) takes the generic family: desc.kind = 4 keeps the handles, the library
% calls them back every step (one MATLAB call per step and handle: rbpf_batch_dyn / feval / rbpf_batch_drn) and does
% weights, normalisation, resampling, Kalman updates and ancestry on the device.  Handles made by rbpf_model carry their
% descriptor and skip all of this.  UNTESTED under MATLAB here (none in the build image); the gateway below this file is
% executed against a mex.h test double (tests/test_gpu_mex_gateway.py).
  if nargin < 4 || isempty(sparseFeatures), sparseFeatures = false; end
  fd = functions(dynModel); fm = functions(measModel);
  ws = first_workspace(fd);
  if isstruct(ws) && isfield(ws, 'rbpf_desc')                         % made by rbpf_model
    desc = ws.rbpf_desc;
    desc.use_dyn_res_norm = ~isempty(dynResNorm);
    return
  end
  desc = propose(fd, fm, sparseFeatures);
  if desc.kind ~= 4
    desc.use_dyn_res_norm = ~isempty(dynResNorm);
    if ~verify(desc, dynModel, measModel, dynResNorm)
      desc = struct('kind', 4);
    end
  end
  if desc.kind == 4
    if sparseFeatures
      error('rbpf:unsupported', 'sparseFeatures = true is implemented for the slam-sparse-visual closures only');
    end
    desc.dynModel = dynModel; desc.measModel = measModel; desc.dynResNorm = dynResNorm;
    desc.use_dyn_res_norm = ~isempty(dynResNorm);
  end
end

function ws = first_workspace(f)
  ws = [];
  if isfield(f, 'workspace') && ~isempty(f.workspace)
    if iscell(f.workspace), ws = f.workspace{1}; else, ws = f.workspace; end
  end
end

function t = squeeze_text(f)
  t = regexprep(f.function, '\s', '');
end

function desc = propose(fd, fm, sparseFeatures)
  desc = struct('kind', 4);
  wd = first_workspace(fd); wm = first_workspace(fm);
  td = squeeze_text(fd); tm = squeeze_text(fm);
  try
    if ~sparseFeatures && ~isempty(regexp(td, 'dynModel$', 'once')) && ~isempty(regexp(tm, 'measModel$', 'once')) ...
        && isstruct(wm) && isfield(wm, 'NN') && size(wm.NN, 2) == 3
      % nested functions of run_dense3D_magfield.m: the parent workspace holds NN and LL (:88-91)
      desc = struct('kind', 1, 'NN', int32(wm.NN), 'L', half_widths(wm, 3));
    elseif ~sparseFeatures && contains(td, 'cos(xn(3))') && contains(td, 'sin(xn(3))') && contains(tm, 'eigenfun(NN,')
      % run_dense2D_withHeading.m:75-76 and :168 (anonymous); measModel captured NN and the eigenfunction handle
      desc = struct('kind', 2, 'NN', int32(wm.NN), 'L', half_widths(wm, 2));
    elseif sparseFeatures && contains(td, 'sqrt(dt*Q)*randn') && contains(tm, 'measurement([xn(1:3);xl]')
      % pfslam.m:81-82 / psslam.m:91-92; measModel captured the camera constants f, fp, fw
      desc = struct('kind', 3, 'nLand', [], 'cam', [wm.f, wm.fp, wm.fw]);
    end
  catch
    desc = struct('kind', 4);
  end
  if isempty(wd), end %#ok<*NOEFF> (dynModel's workspace is not needed: its constants are the arguments)
end

function L = half_widths(ws, d)
  % domain half-widths (tools/domain_cartesian_dx.m:27-29): from LL when the closure sees it, otherwise from the
  % eigenfunction handle, which captured the centred L (:45-50)
  if isfield(ws, 'LL')
    LL = ws.LL;
    if size(LL, 1) > 1, L = (max(LL, [], 1) - min(LL, [], 1)) / 2; else, L = LL; end
  else
    names = {'eigenfun', 'eigenfun_dx'};
    L = [];
    for k = 1:numel(names)
      if isfield(ws, names{k})
        w2 = first_workspace(functions(ws.(names{k})));
        if isstruct(w2) && isfield(w2, 'L'), L = w2.L; break, end
      end
    end
  end
  L = reshape(L(1:d), 1, d);
end

function ok = verify(desc, dynModel, measModel, dynResNorm)
  ok = false;
  try
    if desc.kind == 3
      ok = true; return                       % sparse-visual: nLand is fixed by size(y,2) in the wrapper; text + constants suffice
    end
    if desc.kind == 1, nN = 7; nw = 6; nodo = 7; else, nN = 3; nw = 1; nodo = 3; end
    s0 = rng;                                 % leave the caller's random stream untouched -- also when a handle throws on
    restore = onCleanup(@() rng(s0));         % the probe input (the catch below is reached after randn advanced the stream)
    x = 0.3 * desc.L(1) * randn(nN, 2);
    if desc.kind == 1
      for j = 1:2, x(4:7, j) = x(4:7, j) / norm(x(4:7, j)); end
      dx = [0.1 -0.05 0.02, [1 0.01 -0.02 0.015] / norm([1 0.01 -0.02 0.015])];
      Q = diag([0.25 0.25 0.01 3e-8 3e-8 2.7e-5]); dt = 0.01;
    else
      dx = [0.1 -0.05 0.02]; Q = 0.09; dt = 1;
    end
    a = measModel(x); b = rbpf_mex('measModel', desc, x);
    ok = isequal(size(a), size(b)) && max(abs(a(:) - b(:))) <= 1e-10 * max(1, max(abs(a(:))));
    if ok
      s1 = rng; xa = dynModel(x(:, 1), dx, dt, Q);
      rng(s1);  z = randn(nw, 1);             % the closures draw randn(3,1) twice (mag) / randn once (radio), in this order
      xb = rbpf_mex('dynModel', desc, x(:, 1), dx, dt, Q, z);
      ok = max(abs(xa(:) - xb(:))) <= 1e-10 * max(1, max(abs(xa(:))));
    end
    if ok && ~isempty(dynResNorm)
      ea = dynResNorm(x(:, 1), x(:, 2), dx, dt, Q);
      eb = rbpf_mex('dynResNorm', desc, x(:, 1), x(:, 2), dx, dt, Q);
      ok = numel(ea) == numel(eb) && max(abs(ea(:) - eb(:))) <= 1e-10 * max(1, max(abs(ea(:))));
    end
  catch
    ok = false;
  end
end
