// cloudtrace -- headless equivalent of the reference's entry point for the path-traced image:
// main (src/main.cpp:26-77) -> Tasks::renderCloud (ExecutionLoop/Tasks.cpp:49-112): two tasks, light
// "Side" then "Back", 512x256, 7000 m, output <cloud>.<Light>.PT.exr (Tasks.cpp:88-90; --format pfm for PFM).  The GLUT loop
// (GuiExecutionLoop.cpp:85-128) becomes a plain while(!scene->isCompleted()) scene->update().
//
//   cloudtrace <cloud> [--size WxH] [--spp N] [--mode total|multi|single] [--light Side|Back|Front]
//              [--size-m 7000] [--out DIR] [--data DIR] [--unfused] [--display] [--estimator march|delta] [--format exr|pfm]
//              [--gpus N | --gpus a,b,c]   one process, one shard of 8x8-pixel tiles per GPU, RCCL reduce of [mean | M2] (ct_group_*)
//   <cloud> = file.vdb | procedural:<N>[:<seed>] | file.f32grid
//
//   cloudtrace collect <cloud>|@list.txt [--scene-id I] [--scenes N] [--jobs K] [--gpus ..] [--batch 2048] [--light L] [--size-m M] [--out DIR] [--data DIR] [--estimator ..]
//              = Tasks::collect (Tasks.cpp:114-155) for one SceneSetup: the ScatterSample, Result and DisneyDescriptor
//              collectors one after the other over records [I * batch, (I + 1) * batch), written as flat tables
#include <atomic>
#include <chrono>
#include <cstring>
#include <fstream>
#include <functional>
#include <mutex>
#include <queue>
#include <sstream>
#include <thread>

#include "Cameras.h"
#include "Collectors.h"

using namespace DeepestScatter;

namespace
{
    enum class LightDirection { Front, Side, Back };

    float3 getLightDirection(LightDirection direction)                            // Tasks.cpp:52-65
    {
        switch (direction)
        {
        case LightDirection::Front: return { -0.586f, -0.766f, -0.271f };
        case LightDirection::Side: return { -0.03f, -0.25f, 0.8f };
        case LightDirection::Back: return { 0.586f, -0.766f, -0.271f };
        }
        throw std::invalid_argument("Unexpected direction");
    }

    const char* toString(LightDirection d)
    {
        return d == LightDirection::Front ? "Front" : d == LightDirection::Side ? "Side" : "Back";
    }

    struct Options
    {
        std::string cloud;
        uint32_t width = 512, height = 256;                                       // Tasks.cpp:49-50
        uint32_t spp = 0;                                                         // 0 = until converged
        Cloud::Rendering::Mode mode = Cloud::Rendering::Mode::SunAndSkyAllScatter;
        std::vector<LightDirection> lights = { LightDirection::Side, LightDirection::Back };   // Tasks.cpp:108-109
        float sizeM = 7000.f;                                                     // main.cpp:63
        std::string outDir = ".", dataDir, format = "exr";
        bool fused = true;
        bool display = false;                                                     // --display: tonemap + convergence test after every update, like the GUI
        int estimator = CT_EST_MARCH;                                             // --estimator delta: Woodcock tracking (not the reference's sampler)
        std::vector<int32_t> devices;                                             // --gpus N | --gpus a,b,c: pixel-tile shards, RCCL frame reduce (ct_group_*)
        bool collect = false;                                                     // `cloudtrace collect ...`
        int32_t sceneId = 0;
        uint32_t batch = 2048;                                                    // Tasks.cpp:137
        uint32_t scenes = 1;                                                      // --scenes N: N setups of the same cloud (lights cycle), ids from --scene-id
        uint32_t jobs = 1;                                                        // --jobs K: scene setups in flight at once
    };

    // One line of a scene-setup list = what a Persistance::SceneSetup record carries (cloud path, cloud size, light direction).
    struct SceneSetupLine
    {
        std::string cloud;
        float3 light{};
        float sizeM = 7000.f;
    };

    struct CollectTotals
    {
        double radianceDeviceMs = 0;
        uint64_t experiments = 0;
        uint32_t updates = 0;
    };

    // Tasks::collect (Tasks.cpp:114-155) for one scene setup, with the three collectors the dataset pipeline chains:
    // installSceneSetup(sceneSetup, cloudRoot, SunMultipleScatter, Mipmaps::On), BatchSettings(i * 2048, 2048), EmptyRenderer.
    // Fills `dataset` with the setup's records; returns its timing line.
    std::string collectOne(const Options& opt, const SceneSetupLine& setup, int32_t sceneId, int32_t device, Dataset& out, CollectTotals& totals)
    {
        using Clock = std::chrono::steady_clock;
        const auto msSince = [](Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); };
        auto t0 = Clock::now();
        auto context = std::make_shared<Context>();
        context->devices = { device };
        auto resources = std::make_shared<Resources>(context);
        SceneDescription scene{
            Cloud{ Cloud::Rendering{ 1.0f / 512.f, Cloud::Rendering::Mode::SunMultipleScatter },
                   Cloud::Model{ setup.cloud, Cloud::Model::Mipmaps::On, Cloud::Model::Size{ Meter{ setup.sizeM } } } },
            DirectionalLight{ setup.light, Color{ 1, 1, 1 }, 1e6f } };
        auto sun = std::make_shared<Sun>(std::make_shared<DirectionalLight>(scene.light), context);
        auto cloud = std::make_shared<VDBCloud>(std::make_shared<Cloud::Model>(scene.cloud.model), context, resources);
        auto material = std::make_shared<CloudMaterial>(std::make_shared<Cloud::Rendering>(scene.cloud.rendering), context);
        Resources::loadMie(opt.dataDir, context->mie, context->choppedMie);
        for (const std::shared_ptr<SceneItem>& item : std::vector<std::shared_ptr<SceneItem>>{ sun, cloud, material }) item->init();
        const double loadMs = msSince(t0);                                        // the volume on the host, quantised (Resources.cpp:68-155)
        t0 = Clock::now();
        createCollectorHandle(*context, opt.estimator);
        const double createMs = msSince(t0);                                      // upload, bricks, clearance, shadow volume (VDBCloud::init)

        auto dataset = std::make_shared<Dataset>();
        dataset->batchAppend("SceneSetup", { Persistance::sceneSetup(setup.cloud, setup.sizeM, scene.light.direction.data()) }, sceneId);
        const BatchSettings settings((uint32_t)sceneId * opt.batch, opt.batch);
        auto radiance = std::make_shared<RadianceCollector>(context, dataset, settings);
        std::vector<std::shared_ptr<SceneItem>> collectors{
            std::make_shared<ScatterSampleCollector>(context, dataset, settings, sceneId),
            radiance,
            std::make_shared<DisneyDescriptorCollector>(context, dataset, settings) };
        double phaseMs[3] = { 0, 0, 0 };
        for (size_t k = 0; k < collectors.size(); k++)                            // each is its own task in the reference: init, update until completed
        {
            t0 = Clock::now();
            collectors[k]->init();
            while (!collectors[k]->isCompleted()) collectors[k]->update();
            phaseMs[k] = msSince(t0);
        }
        out.merge(*dataset);
        totals.radianceDeviceMs += radiance->totalRenderMs;
        totals.experiments += radiance->totalExperiments;
        totals.updates += radiance->updates;
        // one line for tools/gpu_collect_bench.sh: where a scene setup's time goes (the reference prints only "MS/Render")
        std::ostringstream line;
        line << "collect_timings {\"scene_id\": " << sceneId << ", \"cloud\": \"" << setup.cloud << "\", \"device\": " << device << ", \"batch\": " << opt.batch
             << ", \"load_ms\": " << loadMs << ", \"create_ms\": " << createMs << ", \"scatter_samples_ms\": " << phaseMs[0] << ", \"radiance_ms\": " << phaseMs[1]
             << ", \"radiance_device_call_ms\": " << radiance->totalRenderMs << ", \"radiance_updates\": " << radiance->updates
             << ", \"radiance_experiments\": " << radiance->totalExperiments << ", \"descriptors_ms\": " << phaseMs[2] << "}";
        return line.str();
    }

    // The scene setups of one run: `collect <cloud>` = one setup from the command line; `collect @list.txt` = one per line,
    // "<cloud> <Side|Back|Front|x,y,z> [sizeM]" (what GenerateSceneSetups.py writes into the SceneSetup table).
    std::vector<SceneSetupLine> readSetups(const Options& opt)
    {
        std::vector<SceneSetupLine> setups;
        if (opt.cloud.empty() || opt.cloud[0] != '@')
        {
            for (uint32_t i = 0; i < std::max(1u, opt.scenes); i++)
                setups.push_back({ opt.cloud, normalize(getLightDirection(opt.lights[i % opt.lights.size()])), opt.sizeM });
            return setups;
        }
        std::ifstream f(opt.cloud.substr(1));
        if (!f) throw std::runtime_error("cannot open scene setup list " + opt.cloud.substr(1));
        std::string row;
        while (std::getline(f, row))
        {
            std::istringstream in(row);
            std::string cloud, light;
            if (!(in >> cloud) || cloud[0] == '#') continue;
            SceneSetupLine s{ cloud, normalize(getLightDirection(LightDirection::Side)), opt.sizeM };
            if (in >> light)
            {
                float3 d{};
                if (std::sscanf(light.c_str(), "%f,%f,%f", &d[0], &d[1], &d[2]) == 3) s.light = normalize(d);
                else s.light = normalize(getLightDirection(light == "Front" ? LightDirection::Front : light == "Back" ? LightDirection::Back : LightDirection::Side));
                in >> s.sizeM;
            }
            setups.push_back(s);
        }
        if (setups.empty()) throw std::runtime_error("no scene setups in " + opt.cloud.substr(1));
        return setups;
    }

    // Tasks::collect's loop over scene setups (Tasks.h:59-71, Tasks.cpp:125-150).  The reference runs them one after the
    // other; a setup's radiance collector is a chain of small dependent launches (an update cannot start before the host has
    // re-packed the tasks the previous one left unconverged) whose length is set by its deepest paths, not by the GPU's
    // throughput, so `--jobs K` keeps K setups in flight -- one host thread and one renderer handle each, their launches
    // overlap on the device -- and `--gpus` deals the setups to the GPUs round-robin (no exchange between them: the setups
    // are independent).  Records do not depend on either: every setup's seeds come from its own record ids.
    int collectScenes(const Options& opt)
    {
        using Clock = std::chrono::steady_clock;
        const std::vector<SceneSetupLine> setups = readSetups(opt);
        const std::vector<int32_t> devices = opt.devices.empty() ? std::vector<int32_t>{ 0 } : opt.devices;
        // with --gpus every listed GPU gets at least one worker (jobs < gpus would leave devices idle), and setup i runs on
        // device i mod gpus whichever worker takes it
        const uint32_t jobs = std::max(1u, std::min<uint32_t>(std::max<uint32_t>(opt.jobs, opt.devices.empty() ? 1u : (uint32_t)devices.size()),
                                                             (uint32_t)setups.size()));
        CollectorLog::quiet() = jobs > 1;
        Dataset dataset;
        CollectTotals totals;
        std::mutex lock;
        std::atomic<uint32_t> next{ 0 };
        std::string firstError;
        const auto t0 = Clock::now();
        const auto worker = [&](uint32_t)
        {
            for (uint32_t i = next++; i < setups.size(); i = next++)
            {
                try
                {
                    Dataset mine;
                    CollectTotals mineTotals;
                    const std::string line = collectOne(opt, setups[i], opt.sceneId + (int32_t)i, devices[i % devices.size()], mine, mineTotals);
                    std::lock_guard<std::mutex> g(lock);
                    dataset.merge(mine);
                    totals.radianceDeviceMs += mineTotals.radianceDeviceMs;
                    totals.experiments += mineTotals.experiments;
                    totals.updates += mineTotals.updates;
                    std::cout << line << std::endl;
                }
                catch (const std::exception& e)
                {
                    std::lock_guard<std::mutex> g(lock);
                    if (firstError.empty()) firstError = e.what();
                    next = (uint32_t)setups.size();                              // stop handing out setups
                }
            }
        };
        std::vector<std::thread> threads;
        for (uint32_t w = 1; w < jobs; w++) threads.emplace_back(worker, w);
        worker(0);
        for (auto& t : threads) t.join();
        if (!firstError.empty()) throw std::runtime_error(firstError);
        const double wallMs = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
        const auto t1 = Clock::now();
        dataset.save(opt.outDir);
        const double saveMs = std::chrono::duration<double, std::milli>(Clock::now() - t1).count();
        std::cout << "wrote " << dataset.getRecordsCount("ScatterSample") << " ScatterSample, " << dataset.getRecordsCount("Result")
                  << " Result, " << dataset.getRecordsCount("DisneyDescriptor") << " DisneyDescriptor records to " << opt.outDir << std::endl;
        std::cout << "collect_totals {\"scene_setups\": " << setups.size() << ", \"jobs\": " << jobs << ", \"gpus\": " << devices.size() << ", \"wall_ms\": " << wallMs
                  << ", \"save_ms\": " << saveMs << ", \"scene_setups_per_hour\": " << setups.size() * 3.6e6 / wallMs << ", \"radiance_updates\": " << totals.updates
                  << ", \"radiance_experiments\": " << totals.experiments << ", \"Mexperiments_per_s\": " << totals.experiments / wallMs / 1e3
                  << ", \"ms_per_update\": " << wallMs / std::max(1u, totals.updates) << "}" << std::endl;
        return 0;
    }

    using LazyTask = std::function<std::shared_ptr<Scene>()>;

    LazyTask renderCloudSingleTask(const Options& opt, LightDirection lightDirection)   // Tasks.cpp:67-102
    {
        return [=]()
        {
            auto context = std::make_shared<Context>();
            context->devices = opt.devices;
            auto resources = std::make_shared<Resources>(context);
            // installSceneSetup (installers.cpp:65-105)
            const float3 direction = normalize(getLightDirection(lightDirection));
            SceneDescription scene{
                Cloud{ Cloud::Rendering{ 1.0f / 512.f, opt.mode },
                       Cloud::Model{ opt.cloud, Cloud::Model::Mipmaps::On, Cloud::Model::Size{ Meter{ opt.sizeM } } } },
                DirectionalLight{ direction, Color{ 1, 1, 1 }, 1e6f } };
            std::string stem = std::filesystem::path(opt.cloud).filename().string();
            std::replace(stem.begin(), stem.end(), ':', '_');
            auto outputPath = std::filesystem::path(opt.outDir) / (stem + "." + toString(lightDirection) + "." + PathTracingRenderer::NAME + "." + opt.format);
            // installFramework + installApp: Sun, VDBCloud, CloudMaterial, Camera in this order (installers.cpp:28-38)
            auto renderer = std::make_shared<PathTracingRenderer>(context);
            renderer->estimator = opt.estimator;
            auto sun = std::make_shared<Sun>(std::make_shared<DirectionalLight>(scene.light), context);
            auto cloud = std::make_shared<VDBCloud>(std::make_shared<Cloud::Model>(scene.cloud.model), context, resources);
            auto material = std::make_shared<CloudMaterial>(std::make_shared<Cloud::Rendering>(scene.cloud.rendering), context);
            auto camera = std::make_shared<Camera>(std::make_shared<Camera::Settings>(opt.width, opt.height, outputPath), context, renderer);
            camera->completed = false;                                            // Tasks.cpp:97-98
            camera->maxSubframes = opt.spp;
            camera->fused = opt.fused;
            camera->headless = opt.fused && !opt.display;
            std::vector<std::shared_ptr<SceneItem>> items{ sun, cloud, material, camera };
            return std::make_shared<Scene>(items, context, opt.dataDir);
        };
    }
}

int main(int argc, char* argv[])
{
    try
    {
        Options opt;
        if (argc < 2) { std::cerr << "usage: cloudtrace <cloud> [--size WxH] [--spp N] [--mode total|multi|single] [--light L] [--size-m M] [--out DIR] [--data DIR] [--unfused] [--display] [--estimator march|delta] [--format exr|pfm] [--gpus N|a,b,c]\n"; return 2; }
        int first = 2;
        opt.cloud = argv[1];
        if (opt.cloud == "collect")
        {
            if (argc < 3) throw std::invalid_argument("cloudtrace collect <cloud> ...");
            opt.collect = true;
            opt.cloud = argv[2];
            opt.lights = { LightDirection::Side };
            first = 3;
        }
        opt.dataDir = (std::filesystem::path(argv[0]).parent_path() / ".." / "data").string();
        for (int i = first; i < argc; i++)
        {
            const std::string a = argv[i];
            auto next = [&]() { if (i + 1 >= argc) throw std::invalid_argument("missing value for " + a); return std::string(argv[++i]); };
            if (a == "--size") { if (std::sscanf(next().c_str(), "%ux%u", &opt.width, &opt.height) != 2) throw std::invalid_argument("--size WxH"); }
            else if (a == "--spp") opt.spp = (uint32_t)std::stoul(next());
            else if (a == "--size-m") opt.sizeM = std::stof(next());
            else if (a == "--out") opt.outDir = next();
            else if (a == "--data") opt.dataDir = next();
            else if (a == "--unfused") opt.fused = false;
            else if (a == "--scene-id") opt.sceneId = std::stoi(next());
            else if (a == "--batch") opt.batch = (uint32_t)std::stoul(next());
            else if (a == "--scenes") opt.scenes = (uint32_t)std::stoul(next());
            else if (a == "--jobs") opt.jobs = (uint32_t)std::stoul(next());
            else if (a == "--gpus")
            {
                // "N" = devices 0..N-1; "a,b,c" = that list (a device may repeat: a rehearsal of N shards on fewer GPUs)
                const std::string v = next();
                opt.devices.clear();
                if (v.find(',') == std::string::npos) { for (int d = 0; d < std::stoi(v); d++) opt.devices.push_back(d); }
                else { size_t p = 0; while (p <= v.size()) { const size_t q = v.find(',', p); opt.devices.push_back(std::stoi(v.substr(p, q - p))); if (q == std::string::npos) break; p = q + 1; } }
                if (opt.devices.empty()) throw std::invalid_argument("--gpus N | --gpus a,b,c");
            }
            else if (a == "--display") opt.display = true;
            else if (a == "--estimator")
            {
                const std::string e = next();
                if (e == "march") opt.estimator = CT_EST_MARCH;
                else if (e == "delta") opt.estimator = CT_EST_DELTA;
                else throw std::invalid_argument("--estimator march|delta");
            }
            else if (a == "--format") { opt.format = next(); if (opt.format != "exr" && opt.format != "pfm") throw std::invalid_argument("--format exr|pfm"); }
            else if (a == "--mode")
            {
                const std::string m = next();
                if (m == "total") opt.mode = Cloud::Rendering::Mode::SunAndSkyAllScatter;
                else if (m == "multi") opt.mode = Cloud::Rendering::Mode::SunMultipleScatter;
                else if (m == "single") opt.mode = Cloud::Rendering::Mode::SunSingleScatter;
                else throw std::invalid_argument("Invalid Render Mode");           // CloudMaterial.cpp:62
            }
            else if (a == "--light")
            {
                const std::string l = next();
                opt.lights = { l == "Front" ? LightDirection::Front : l == "Back" ? LightDirection::Back : LightDirection::Side };
            }
            else throw std::invalid_argument("unknown option " + a);
        }

        if (opt.collect) return collectScenes(opt);

        std::queue<LazyTask> tasks;                                               // Tasks::renderCloud, Tasks.cpp:104-112
        for (auto l : opt.lights) tasks.push(renderCloudSingleTask(opt, l));

        while (!tasks.empty())                                                    // GuiExecutionLoop::getNextTask
        {
            auto scene = tasks.front()();
            tasks.pop();
            scene->init();
            while (!scene->isCompleted())                                         // glutDisplay, GuiExecutionLoop.cpp:114-128
            {
                const auto t1 = std::chrono::steady_clock::now();
                scene->update();
                const auto t2 = std::chrono::steady_clock::now();
                std::cout << "MS/FRAME " << std::chrono::duration<double, std::milli>(t2 - t1).count() << std::endl;
            }
        }
        return 0;
    }
    catch (const std::exception& e)                                               // main.cpp:65-76
    {
        std::cout << e.what() << std::endl;
        return 1;
    }
}
